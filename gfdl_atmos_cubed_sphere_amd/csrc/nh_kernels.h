// nh_kernels.h -- the nonhydrostatic column path of the acoustic substep.
//
//   UpdateDzC     update_dz_c      model/nh_utils.F90:59-201     (column kernel)
//   RiemSolverC   Riem_Solver_c    model/nh_utils.F90:323-480    (column kernel, SIM1_solver :1277-1394)
//   EdgeProfile   edge_profile     model/nh_utils.F90:1590-1696  (column kernel, part of update_dz_d)
//   ZhTransport   update_dz_d      model/nh_utils.F90:256-301    (tile kernel: fv_tp_2d per interface)
//   ZhLimit       update_dz_d      model/nh_utils.F90:303-319    (column kernel)
//   RiemSolver3   Riem_Solver3     model/nh_core.F90:47-241      (column kernel, SIM1/SIM_solver)
//   PGradC        p_grad_c         model/dyn_core.F90:1635-1694  (pointwise)
//   A2BCorners/NhPGrad  nh_p_grad  model/dyn_core.F90:1697-1792  (tile kernel + pointwise)
//   Pk3Halo/PeHalo pk3_halo, pln_halo, pe_halo  model/dyn_core.F90:1395-1526 (strip columns)
//   Geopk         geopk            model/dyn_core.F90:2202-2353  (column kernel)
//
// Column kernels: one thread per (i,j) column, consecutive threads = consecutive i (coalesced
// 512-B rows per wavefront at every level), k sequential.  The tridiagonal sweeps keep their
// O(km) intermediates in context-owned scratch slabs laid out like the fields (i fastest), so every
// scratch access is coalesced too.  Branches: use_cond = moist_kappa = .false., fast_tau_w_sec = 0,
// d2bg_zq = 0.
#pragma once

#include "fv3_common.h"
#include "cubed_common.h"
#include "remap_kernels.h"  // RemapPar, moist_cv
#include "tp2d_tile.h"

namespace fv3 {

constexpr double kDzMin = 2.;  // nh_utils.F90:49

struct NhConsts {
  double grav, rdgas, cp_air, akap, ptop, p_fac, a_imp;
  // fast_tau_w_sec > 0 (nh_utils.F90:356-367, :1363-1371, :1498-1506): rff(k) of the Rayleigh damping of w inside SIM1_solver / SIM_solver,
  // npz values on the device, 1.0 below k_rf (x * 1.0 is x: the reference's loop over k <= k_rf); null = off (fv3_set_fast_tau_w)
  const double *rff = nullptr;
};

#define FV3_COL_FOR(c, ncol) for (int c = bx * 256 + tid; c < (bx + 1) * 256 && c < (ncol); c += kNT)
// The same with a POOL of P workgroups (P > 0: the launch has P workgroups, each takes the column blocks bx, bx + P, ...; cs is
// the column slot of the scratch slabs: a workgroup reuses its own 256 slots for every block it takes, so the slabs of a launch
// are P x 256 columns that stay in L2 / Infinity Cache instead of one pass through HBM per sweep of the solver).  P = 0: one
// workgroup per column block, cs = c.
#define FV3_COL_FOR_POOL(c, cs, ncol, P)                                            \
  for (int bb_ = bx; bb_ < ((ncol) + 255) / 256; bb_ += ((P) > 0 ? (P) : 0x40000000)) \
    for (int c = bb_ * 256 + tid, cs = ((P) > 0 ? bx : bb_) * 256 + tid; c < (bb_ + 1) * 256 && c < (ncol); c += kNT, cs += kNT)

// ------------------------------------------------------------------------------------------------
#ifdef FV3_HOST_EMU
#define FV3_RESTRICT
#define FV3_UNROLL4
#define FV3_UNROLL_ALL
#else
#define FV3_RESTRICT __restrict__
#define FV3_UNROLL4 _Pragma("unroll 4")
#define FV3_UNROLL_ALL _Pragma("unroll")
#endif
// One sweep from the surface up: the flux-form update of a level does not depend on the other levels, and the monotonicity fix
// (:193-199) runs from the bottom, so the thread applies it to each level as it is formed and writes gz once (the first form of this
// kernel swept down forming gz and up again fixing it: 16 word accesses per level, 8 of them the winds of a level read for both of
// its interfaces; here 4 + the 5 heights + 1 store).  The interface weights of dp_ref come from an LDS table: as global loads they
// cannot be scalar (the kernel stores to global memory) and would queue with the field loads.  Values are bit for bit those of the
// two-sweep form.
struct UpdateDzC {
  Grid g;
  int km;
  double dt;
  const double *dp0;  // device, km
  const double *zs, *ut, *vt, *gz_in;
  double *gz, *ws;
  static size_t lds_doubles(int km) { return 3 * (size_t)(km + 2); }
  FV3_HD void operator()(int bx, int, int, int tid, double *lds) const {
    const int w = g.nx + 2, ncol = w * (g.ny + 2);
    const size_t nA = g.nA();
    const double rdt = 1. / dt;
    const double top_ratio = dp0[0] / (dp0[0] + dp0[1]);
    const double bot_ratio = dp0[km - 1] / (dp0[km - 2] + dp0[km - 1]);
    double *t_wa = lds, *t_wb = lds + km + 2, *t_ir = lds + 2 * (km + 2);   // interface k = 2 .. km: dp0(k), dp0(k-1), 1 / (dp0(k-1) + dp0(k))
    for (int k = tid; k < km + 2; k += kNT) {
      const bool in = k >= 2 && k <= km;
      t_wa[k] = in ? dp0[k - 1] : 0.;
      t_wb[k] = in ? dp0[k - 2] : 0.;
      t_ir[k] = in ? 1. / (dp0[k - 2] + dp0[k - 1]) : 0.;
    }
    FV3_SYNC();
    const double *FV3_RESTRICT U = ut, *FV3_RESTRICT V = vt, *FV3_RESTRICT Z = gz_in;
    double *FV3_RESTRICT G = gz;
    FV3_COL_FOR(c, ncol) {
      const int i = g.is - 1 + c % w, j = g.js - 1 + c / w;
      const int o = g.iA(i, j);
      int oe = g.iA(i + 1, j), on = g.iA(i, j + 1), ow = g.iA(i - 1, j), os = g.iA(i, j - 1), cx_ = o, cy_ = o;
      // flux-velocity rows are read at the true neighbours (oe, on below are re-used for the gz reads after the map)
      const int oe_v = oe, on_v = on;
      if (g.grid_type < 3) {
        auto m = [&](int dir, int ii, int jj) { fill4_src(dir, g.npx, g.npy, ii, jj); return g.iA(ii, jj); };
        ow = m(1, i - 1, j); oe = m(1, i + 1, j); cx_ = m(1, i, j);
        os = m(2, i, j - 1); on = m(2, i, j + 1); cy_ = m(2, i, j);
      }
      const double ar = g.area[o];
      // cubed sphere: fill_4corners(gz2, 1) before the x fluxes, (gz2, 2) before the y fluxes (nh_utils.F90:151,163), as
      // index maps on the reads; the cell value of the update is what the second fill left (identity off the corners)
      struct Z6 { double cx, cy, w, e, s, n; };   // the heights of one interface around the cell: all six loaded, then the upwind choice
      struct W4 { double uo, ue, vo, vn; };       // winds of one layer at the cell and at its east / north neighbour
      auto heights = [&](int k) {
        const double *z = Z + (size_t)((k > 1 ? k : 1) - 1) * nA;
        return Z6{z[cx_], z[cy_], z[ow], z[oe], z[os], z[on]};
      };
      auto winds = [&](int l) {
        const size_t b = (size_t)(l > 0 ? l : 0) * nA;
        return W4{U[b + o], U[b + oe_v], V[b + o], V[b + on_v]};
      };
      auto height = [&](const Z6 &z, double x0, double x1, double y0, double y1) {
        const double fx0 = x0 * ((x0 > 0.) ? z.w : z.cx), fx1 = x1 * ((x1 > 0.) ? z.cx : z.e);
        const double fy0 = y0 * ((y0 > 0.) ? z.s : z.cy), fy1 = y1 * ((y1 > 0.) ? z.cy : z.n);
        return (z.cy * ar + fx0 - fx1 + fy0 - fy1) / (ar + x0 - x1 + y0 - y1);
      };
      // The loads of kDep interfaces ahead are in flight while one is worked on: slot s holds the heights of the interface the sweep
      // reaches s steps from now and the winds of the layer that enters with it; a slot is refilled as soon as it is used (clamped
      // addresses, no branch around a load).
      constexpr int kDep = 4;
      Z6 zb[kDep];
      W4 wb[kDep];
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++) {
        zb[s] = heights(km - s);
        wb[s] = winds(km - s - 3);
      }
      W4 B = winds(km - 1), A = winds(km - 2);   // interface k between layers k-1 and k (1-based): A = layer k-1, B = layer k
      double below = height(heights(km + 1), B.uo + (B.uo - A.uo) * bot_ratio, B.ue + (B.ue - A.ue) * bot_ratio,
                            B.vo + (B.vo - A.vo) * bot_ratio, B.vn + (B.vn - A.vn) * bot_ratio);
      G[(size_t)km * nA + o] = below;
      ws[o] = (zs[o] - below) * rdt;
      auto level = [&](int k, const Z6 &z, const W4 &wn) {
        double x0, x1, y0, y1;
        if (k > 1) {
          const double wa = t_wa[k], wb_ = t_wb[k], ir = t_ir[k];
          x0 = (wa * A.uo + wb_ * B.uo) * ir; x1 = (wa * A.ue + wb_ * B.ue) * ir;
          y0 = (wa * A.vo + wb_ * B.vo) * ir; y1 = (wa * A.vn + wb_ * B.vn) * ir;
        } else {                                  // A = layer 1, B = layer 2
          x0 = A.uo + (A.uo - B.uo) * top_ratio; x1 = A.ue + (A.ue - B.ue) * top_ratio;
          y0 = A.vo + (A.vo - B.vo) * top_ratio; y1 = A.vn + (A.vn - B.vn) * top_ratio;
        }
        const double v = dmax(height(z, x0, x1, y0, y1), below + kDzMin);
        G[(size_t)(k - 1) * nA + o] = v;
        below = v;
        if (k >= 3) {                             // the layers of interface k - 1 (interface 2 leaves them to interface 1)
          B = A;
          A = wn;
        }
      };
      int kk = km;
      for (; kk - kDep + 1 >= 1; kk -= kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const int k = kk - s;
          const Z6 z = zb[s];
          const W4 wn = wb[s];
          zb[s] = heights(k - kDep);
          wb[s] = winds(k - kDep - 3);
          level(k, z, wn);
        }
      }
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++)
        if (kk - s >= 1) level(kk - s, zb[s], wb[s]);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The semi-implicit solver for one column (SIM1_solver nh_utils.F90:1277-1394 when sim1, SIM_solver
// :1396-1537 otherwise).  Column data are addressed as base[(k-1)*ls] (k = 1..km).  Scratch slabs
// s_gam, s_pp, s_w, s_pm (each (km+1) levels, same addressing).  On exit: s_pp[k] = pe2(k) (k=1..km+1)
// -- the nonhydrostatic pressure perturbation -- s_w[k] = w2(k), and dz2 is returned through the
// callback-free convention: s_gam[k] = dz2(k) (k = 1..km).
struct ColIn {
  const double *delp, *pt, *w, *zlev;  // zlev: gz (C) or zh (D) interface heights, km+1 levels
  double zscale;                       // dz2 = (zlev(k+1)-zlev(k)) * zscale   (1 for both; kept for clarity)
  // MOIST columns only (same level stride): q_con (use_cond: the hydrostatic pressure of pm2 excludes the condensates,
  // nh_core.F90:113-131,145-154 / nh_utils.F90:383-396,413-438) and cappa (moist_kappa: per-cell kappa) or null
  const double *qcon = nullptr, *cappa = nullptr;
};

// The scratch slabs never alias the inputs or each other: with that stated (and the k loops unrolled by 4) the
// compiler issues the loads of the next levels ahead of the dependent recurrence instead of one level at a time.
// Results are handed to the caller's sinks while the last two sweeps run (no extra passes over the scratch slabs):
//   on_pe(k, pe2(k)), k = 1..km+1 ascending  -- the nonhydrostatic pressure perturbation at the interfaces,
//   on_w(k, w2(k)),  k = 1..km               -- the new vertical velocity (may overwrite in.w: level k is not read again),
//   on_dz(k, dz2(k)), k = km..1 descending   -- the new layer thickness (may overwrite in.zlev: not read after pass C).
template <bool MOIST = false, class OnPe, class OnW, class OnDz>
FV3_HD void sim_column(int km, size_t ls, size_t ss, const ColIn &in, double dt, const NhConsts &cn, bool sim1, bool c_grid,
                       double ws, double *FV3_RESTRICT s_gam, double *FV3_RESTRICT s_pp, double *FV3_RESTRICT s_w,
                       double *FV3_RESTRICT s_pm, const OnPe &on_pe, const OnW &on_w, const OnDz &on_dz) {
  constexpr double r3 = 1. / 3.;
  const double rgrav = 1. / cn.grav, rgas = cn.rdgas;
  const double gm2c = 1. / (1. - cn.akap), cp2c = cn.akap;
  // gm2(k), cp2(k): constants unless the column carries cappa (C grid: only together with q_con, nh_utils.F90:413-425)
  const bool has_cappa = MOIST && in.cappa && (!c_grid || in.qcon);
  auto cp2_at = [&](int k) { return has_cappa ? in.cappa[(size_t)(k - 1) * ls] : cp2c; };
  auto gm2_at = [&](int k) { return has_cappa ? 1. / (1. - in.cappa[(size_t)(k - 1) * ls]) : gm2c; };
  const double alpha = cn.a_imp, beta = 1. - alpha, ra = 1. / alpha, t2 = beta / alpha;
  const double t1g = sim1 ? 2. * dt * dt : 2. * ((alpha * dt) * (alpha * dt));
  const double rdt = 1. / dt;
#define L(p, k) (p)[(size_t)((k)-1) * ls]
#define S(p, k) (p)[(size_t)((k)-1) * ss]   // scratch slabs: level stride ss (64 in per-wavefront blocks, see scr_col)
  // ---- pass A: pe(k), pm2(k); forward elimination for pp (:1297-1326) ----
  double pem_k = cn.ptop, peln_k = dlog(cn.ptop);
  double peg_k = cn.ptop, pelng_k = peln_k;  // MOIST + q_con: dry-gas + vapour hydrostatic pressure and its log
  double z_top = L(in.zlev, 1);  // zlev(k) of the level being set up: each interface height is loaded once
  auto level = [&](int k, double &dm2, double &dz2, double &pm2, double &pe, double &pem_next, double &peln_next) {
    const double dmr = L(in.delp, k);
    pem_next = pem_k + dmr;
    if (c_grid) {
      pm2 = dmr / dlog(pem_next / pem_k);  // nh_utils.F90:440
      peln_next = 0.;
    } else {
      peln_next = dlog(pem_next);          // nh_core.F90:140,159
      pm2 = dmr / (peln_next - peln_k);
    }
    if (MOIST && in.qcon) {               // excluding the contribution from condensates
      const double peg_next = peg_k + dmr * (1. - L(in.qcon, k));
      if (c_grid) {
        pm2 = (peg_next - peg_k) / dlog(peg_next / peg_k);      // nh_utils.F90:418,429
      } else {
        const double pelng_next = dlog(peg_next);               // nh_core.F90:126-127
        pm2 = (peg_next - peg_k) / (pelng_next - pelng_k);     // :148
        pelng_k = pelng_next;
      }
      peg_k = peg_next;
    }
    dm2 = dmr * rgrav;
    const double z_bot = L(in.zlev, k + 1);
    dz2 = z_bot - z_top;
    z_top = z_bot;
    pe = dexp(gm2_at(k) * dlog(-dm2 / dz2 * rgas * L(in.pt, k))) - pm2;
  };
  double dm_c, dz_c, pm_c, pe_c, pem_n, peln_n;
  level(1, dm_c, dz_c, pm_c, pe_c, pem_n, peln_n);
  S(s_pm, 1) = pm_c;
  double bet = 0., rbet = 0., pp_k = 0., g_rat_prev = 0.;
  S(s_pp, 1) = 0.;
  FV3_UNROLL4
  for (int k = 1; k <= km; k++) {
    double dm_n = 0., dz_n = 0., pm_n = 0., pe_n = 0., pem_nn = 0., peln_nn = 0.;
    double bb, dd, g_rat = 0.;
    if (k < km) {
      pem_k = pem_n;
      peln_k = peln_n;
      level(k + 1, dm_n, dz_n, pm_n, pe_n, pem_nn, peln_nn);
      S(s_pm, k + 1) = pm_n;
      g_rat = dm_c / dm_n;
      bb = 2. * (1. + g_rat);
      dd = 3. * (pe_c + g_rat * pe_n);
    } else {
      bb = 2.;
      dd = 3. * pe_c;
    }
    // every bet divides two numerators (pp of this level, g_rat of the next): one reciprocal, the quotients through it
    // (correctly rounded: remap_kernels.h rcp_rn / div_rn)
    if (k == 1) {
      bet = bb;
      rbet = rcp_rn(bet);
      pp_k = div_rn(dd, bet, rbet);  // pp(2)
    } else {
      const double gam = div_rn(g_rat_prev, bet, rbet);
      bet = bb - gam;
      rbet = rcp_rn(bet);
      S(s_gam, k) = gam;
      pp_k = div_rn(dd - pp_k, bet, rbet);  // pp(k+1)
    }
    S(s_pp, k + 1) = pp_k;
    g_rat_prev = g_rat;
    dm_c = dm_n; dz_c = dz_n; pm_c = pm_n; pe_c = pe_n; pem_n = pem_nn; peln_n = peln_nn;
  }
  // ---- pass B: back substitution (:1328-1332) ----
  {
    double pp_next = S(s_pp, km + 1);
    FV3_UNROLL4
    for (int k = km; k >= 2; k--) {
      const double v = S(s_pp, k) - S(s_gam, k) * pp_next;
      S(s_pp, k) = v;
      pp_next = v;
    }
  }
  // ---- pass C: forward sweep of the w solver (:1335-1356 / :1463-1491) ----
  {
    double pem = cn.ptop;                       // pem(k)
    double dz_prev = 0., w_prev = 0., w1_prev = 0., aa_k = 0., wk_k = 0.;
    double dm1 = 0.;
    // every interface height, pp and w value is loaded once and carried to the next level
    double z_lo = L(in.zlev, 2);
    double dz_c2 = z_lo - L(in.zlev, 1);
    double pp_lo = S(s_pp, 1), w_c = L(in.w, 1);
    FV3_UNROLL4
    for (int k = 1; k <= km; k++) {
      const double dmr = L(in.delp, k), dm2 = dmr * rgrav;
      const double dz2 = dz_c2;
      const double w1 = w_c;
      const double pp_k0 = pp_lo, pp_k1 = S(s_pp, k + 1);
      pp_lo = pp_k1;
      if (k == 1) dm1 = dm2;
      // aa(k+1), wk(k+1) need level k+1
      double aa_n = 0., wk_n = 0., pem_next = pem + dmr;
      if (k < km) {
        const double z_n = L(in.zlev, k + 2);
        const double dz_n = z_n - z_lo;
        z_lo = z_n;
        dz_c2 = dz_n;
        w_c = L(in.w, k + 1);
        aa_n = t1g * 0.5 * (gm2_at(k) + gm2_at(k + 1)) / (dz2 + dz_n) * pem_next;
        if (!sim1) {
          wk_n = t2 * aa_n * (w1 - w_c);
          aa_n = aa_n - 0.0 * dm1;  // scale_m = 0 (nh_utils.F90:1467)
        }
      }
      double w2;
      if (k == 1) {
        bet = dm2 - aa_n;
        rbet = rcp_rn(bet);
        w2 = sim1 ? div_rn(dm2 * w1 + dt * pp_k1, bet, rbet) : div_rn(dm2 * w1 + dt * pp_k1 + wk_n, bet, rbet);
      } else if (k < km) {
        const double gam = div_rn(aa_k, bet, rbet);
        bet = dm2 - (aa_k + aa_n + aa_k * gam);
        rbet = rcp_rn(bet);
        S(s_gam, k) = gam;
        w2 = sim1 ? div_rn(dm2 * w1 + dt * (pp_k1 - pp_k0) - aa_k * w_prev, bet, rbet)
                  : div_rn(dm2 * w1 + dt * (pp_k1 - pp_k0) + wk_n - wk_k - aa_k * w_prev, bet, rbet);
      } else {
        const double p1 = t1g * gm2_at(km) / dz2 * pem_next;  // pem(km+1)
        const double gam = div_rn(aa_k, bet, rbet);
        bet = dm2 - (aa_k + p1 + aa_k * gam);
        rbet = rcp_rn(bet);
        S(s_gam, k) = gam;
        w2 = sim1 ? div_rn(dm2 * w1 + dt * (pp_k1 - pp_k0) - p1 * ws - aa_k * w_prev, bet, rbet)
                  : div_rn(dm2 * w1 + dt * (pp_k1 - pp_k0) - wk_k + p1 * (t2 * w1 - ra * ws) - aa_k * w_prev, bet, rbet);
      }
      S(s_w, k) = w2;
      w_prev = w2;
      w1_prev = w1;
      dz_prev = dz2;
      aa_k = aa_n;
      wk_k = wk_n;
      pem = pem_next;
    }
    (void)w1_prev; (void)dz_prev;
  }
  // ---- pass D: back substitution for w (:1357-1361) ----
  {
    double w_next = S(s_w, km);
    FV3_UNROLL4
    for (int k = km - 1; k >= 1; k--) {
      const double v = S(s_w, k) - S(s_gam, k + 1) * w_next;
      S(s_w, k) = v;
      w_next = v;
    }
  }
  // ---- pass E: pe(k+1) = pe(k) + dm2*(w2-w1)*rdt (:1373-1380 / :1508-1516); pp kept for the SIM blend ----
  {
    double pe = 0.;
    double pp_k2 = S(s_pp, 1);
    // s_gam is free now: keep pp there for the final blend of SIM_solver (:1531-1535)
    FV3_UNROLL4
    for (int k = 1; k <= km; k++) {
      const double dm2 = L(in.delp, k) * rgrav;
      const double pp_n = S(s_pp, k + 1);
      S(s_pp, k) = pe;
      if (!sim1) S(s_gam, k) = pp_k2;
      double w2 = S(s_w, k);
      const double w1 = L(in.w, k);
      if (cn.rff) w2 = w2 * cn.rff[k - 1];   // :1363-1371 / :1498-1506, after the back substitution
      if (sim1) {
        on_pe(k, pe);      // SIM1: pe2 is final here (no blend)
        pe = pe + dm2 * (w2 - w1) * rdt;
      } else {
        pe = pe + (dm2 * (w2 - w1) * rdt - beta * (pp_n - pp_k2)) * ra;
      }
      on_w(k, w2);
      pp_k2 = pp_n;
    }
    S(s_pp, km + 1) = pe;
    if (sim1) on_pe(km + 1, pe);
    if (!sim1) S(s_gam, km + 1) = pp_k2;
  }
  // ---- pass F: new layer thickness (:1382-1392 / :1518-1529); dz2 -> s_pm (pm2 consumed level by level)
  {
    double pp_1 = S(s_pp, km), pp_2 = S(s_pp, km + 1), pp_0 = 0.;  // pe2(k+1), pe2(k+2) carried downwards
    double p1 = (pp_1 + 2. * pp_2) * r3;
    double dm_below = 0.;
    FV3_UNROLL4
    for (int k = km; k >= 1; k--) {
      const double dm2 = L(in.delp, k) * rgrav, pm2 = S(s_pm, k);
      if (k < km) {
        pp_0 = S(s_pp, k);
        const double g_rat = dm2 / dm_below, bb = 2. * (1. + g_rat);
        p1 = (pp_0 + bb * pp_1 + g_rat * pp_2) * r3 - g_rat * p1;
        pp_2 = pp_1;
        pp_1 = pp_0;
      }
      on_dz(k, -dm2 * rgas * L(in.pt, k) * dexp((cp2_at(k) - 1.) * dlog(dmax(cn.p_fac * pm2, p1 + pm2))));
      dm_below = dm2;
    }
  }
  if (!sim1) {  // pe2 = pe2 + beta*(pp - pe2) (:1531-1535)
    for (int k = 1; k <= km + 1; k++) on_pe(k, S(s_pp, k) + beta * (S(s_gam, k) - S(s_pp, k)));
  }
#undef S
#undef L
}

// MOIST is a launch-time choice (separate kernels): the dry kernel must not carry the registers of the moist branches
template <bool MOIST>
struct RiemSolverC {
  Grid g;
  int km;
  double dt;
  NhConsts cn;
  const double *hs, *w3, *pt, *delp, *ws;
  double *gz, *pef;
  double *s0, *s1, *s2, *s3;  // scratch slabs, A x (km+1)
  const double *q_con, *cappa;  // A x km or null (use_cond / moist_kappa)
  int scr_blocked;              // scratch slabs in per-wavefront blocks (remap_kernels.h scr_col)
  int pool;                     // workgroups of a pooled launch (FV3_COL_FOR_POOL; needs scr_blocked), 0 = one per column block
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int w = g.nx + 2, ncol = w * (g.ny + 2);
    const size_t nA = g.nA();
    FV3_COL_FOR_POOL(c, cs, ncol, pool) {
      const int i = g.is - 1 + c % w, j = g.js - 1 + c / w;
      const int o = g.iA(i, j);
      ColIn in{delp + o, pt + o, w3 + o, gz + o, 1.};
      if (MOIST) {
        in.qcon = q_con + o;
        in.cappa = cappa ? cappa + o : nullptr;
      }
      // pef = pe2 + pem (:461-465); gz = hs - sum dz2*grav (:468-476), formed inside the solver's last two sweeps
      double pem = cn.ptop;
      double zb = hs[o];
      const size_t so = scr_blocked ? (size_t)(cs >> 6) * 64 * (km + 1) + (cs & 63) : (size_t)o;
      const size_t ss = scr_blocked ? 64 : nA;
      sim_column<MOIST>(
          km, nA, ss, in, dt, cn, true, true, ws[o], s0 + so, s1 + so, s2 + so, s3 + so,
          [&](int k, double pe2) {
            if (k == 1) {
              pef[o] = cn.ptop;
            } else {
              pem = pem + delp[(size_t)(k - 2) * nA + o];
              pef[(size_t)(k - 1) * nA + o] = pe2 + pem;
            }
          },
          [&](int, double) {},
          [&](int k, double dz2) {
            if (k == km) gz[(size_t)km * nA + o] = zb;
            zb = zb - dz2 * cn.grav;
            gz[(size_t)(k - 1) * nA + o] = zb;
          });
    }
  }
};

template <bool MOIST>
struct RiemSolver3 {
  Grid g;
  int km;
  double dt;
  NhConsts cn;
  const double *zs, *pt, *delp, *ws;
  double *w, *delz, *zh, *pe, *ppe, *pk3, *pk, *peln;
  int use_logp, last_call, fp_out;
  double *s0, *s1, *s2, *s3;
  const double *q_con, *cappa;  // A x km or null (use_cond / moist_kappa, nh_core.F90:96-166)
  int scr_blocked;
  int pool;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    const bool sim1 = cn.a_imp > 0.999;
    const double peln1 = dlog(cn.ptop), ptk = dexp(cn.akap * peln1);
    FV3_COL_FOR_POOL(c, cs, ncol, pool) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      ColIn in{delp + o, pt + o, w + o, zh + o, 1.};
      if (MOIST) {
        in.qcon = q_con ? q_con + o : nullptr;
        in.cappa = cappa ? cappa + o : nullptr;
      }
      // hydrostatic pressure functions (:132-143) and the outputs (:191-237) are formed inside the solver's last sweeps
      double pem = cn.ptop;
      double zb = zs[o];
      const size_t so = scr_blocked ? (size_t)(cs >> 6) * 64 * (km + 1) + (cs & 63) : (size_t)o;
      const size_t ss = scr_blocked ? 64 : nA;
      sim_column<MOIST>(
          km, nA, ss, in, dt, cn, sim1, false, ws[occ], s0 + so, s1 + so, s2 + so, s3 + so,
          [&](int k, double pe2) {
            if (k == 1) {
              pk3[o] = ptk;
              if (last_call) {
                peln[(size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is)] = peln1;
                pk[occ] = ptk;
                pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1))] = cn.ptop;
              }
              ppe[o] = fp_out ? pe2 + pem : pe2;
              return;
            }
            pem = pem + delp[(size_t)(k - 2) * nA + o];
            // (log(pem) is also formed in the solver's first sweep; passing it through pk3 instead of recomputing it was
            // measured: the extra store and load cost more than the log, 12.4 -> 13.5 ms per dt_atmos)
            const double pl = dlog(pem), pkv = dexp(cn.akap * pl);
            pk3[(size_t)(k - 1) * nA + o] = use_logp ? pl : pkv;
            if (last_call) {
              peln[(size_t)(j - g.js) * g.nx * (km + 1) + (size_t)(k - 1) * g.nx + (i - g.is)] = pl;
              pk[(size_t)(k - 1) * nCC + occ] = pkv;
              pe[(size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (size_t)(k - 1) * (g.nx + 2) + (i - (g.is - 1))] = pem;
            }
            ppe[(size_t)(k - 1) * nA + o] = fp_out ? pe2 + pem : pe2;
          },
          [&](int k, double w2) { w[(size_t)(k - 1) * nA + o] = w2; },
          [&](int k, double dz) {
            if (k == km) zh[(size_t)km * nA + o] = zb;
            delz[(size_t)(k - 1) * nCC + occ] = dz;
            zb = zb - dz;
            zh[(size_t)(k - 1) * nA + o] = zb;
          });
    }
  }
};

// ------------------------------------------------------------------------------------------------
// edge_profile (non-uniform branch, limiter=0).  gk[k], bet[k], gam[k] (k = 1..km, index k-1) and the
// end coefficients depend only on dp0 and are precomputed on the host with the reference's arithmetic.
struct EdgeCoef {
  const double *gk, *bet, *gam;  // device, km each (gk[0] unused)
  double xt1_top, bet_top, xt1_bot, a_bot, gk_bot;
};

struct EdgeProfile {
  Grid g;
  int km;
  EdgeCoef ec;
  const double *q1, *q2;  // km levels
  double *q1e, *q2e;      // km+1 levels
  int n2d;                // points per level (nCX or nCY)
  // a second pair of fields with its own level size (update_dz_d: crx / xfx on CX and cry / yfx on CY in ONE launch -- twice the
  // wavefronts for a kernel whose time is the dependent chain of a division per level); n2d_b = 0: none
  const double *q1_b = nullptr, *q2_b = nullptr;
  double *q1e_b = nullptr, *q2e_b = nullptr;
  int n2d_b = 0;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    FV3_COL_FOR(cc, n2d + n2d_b) {
      const bool second = cc >= n2d;
      const int c = second ? cc - n2d : cc;
      const size_t ls = (size_t)(second ? n2d_b : n2d);
      const double *q1 = second ? this->q1_b : this->q1, *q2 = second ? this->q2_b : this->q2;
      double *q1e = second ? this->q1e_b : this->q1e, *q2e = second ? this->q2e_b : this->q2e;
      double a_prev = q1[c], b_prev = q2[c];
      double a_cur = q1[ls + c], b_cur = q2[ls + c];
      double e1 = (ec.xt1_top * a_prev + a_cur) / ec.bet_top, e2 = (ec.xt1_top * b_prev + b_cur) / ec.bet_top;
      q1e[c] = e1;
      q2e[c] = e2;
      for (int k = 2; k <= km; k++) {
        a_cur = q1[(size_t)(k - 1) * ls + c];
        b_cur = q2[(size_t)(k - 1) * ls + c];
        const double gk = ec.gk[k - 1], bet = ec.bet[k - 1];
        e1 = (3. * (a_prev + gk * a_cur) - e1) / bet;
        e2 = (3. * (b_prev + gk * b_cur) - e2) / bet;
        q1e[(size_t)(k - 1) * ls + c] = e1;
        q2e[(size_t)(k - 1) * ls + c] = e2;
        if (k < km) {
          a_prev = a_cur;
          b_prev = b_cur;
        }
      }
      // a_prev = q(km-1), a_cur = q(km)
      const double xt2 = ec.gk_bot * (ec.gk_bot + 0.5) - ec.a_bot * ec.gam[km - 1];
      e1 = (ec.xt1_bot * a_cur + a_prev - ec.a_bot * e1) / xt2;
      e2 = (ec.xt1_bot * b_cur + b_prev - ec.a_bot * e2) / xt2;
      q1e[(size_t)km * ls + c] = e1;
      q2e[(size_t)km * ls + c] = e2;
      for (int k = km; k >= 1; k--) {
        e1 = q1e[(size_t)(k - 1) * ls + c] - ec.gam[k - 1] * e1;
        e2 = q2e[(size_t)(k - 1) * ls + c] - ec.gam[k - 1] * e2;
        q1e[(size_t)(k - 1) * ls + c] = e1;
        q2e[(size_t)(k - 1) * ls + c] = e2;
      }
    }
  }
};

// fv_tp_2d of every interface height + the flux-form update (update_dz_d :256-301)
template <int TI, int TJ>
struct ZhTransport {
  Grid g;
  int hord;
  const double *zh, *crx, *cry, *xfx, *yfx;  // interface-level Courant numbers / fluxes (km+1 levels)
  const int *ndif;                            // device, km+1
  const double *damp;                         // device, km+1
  double *zh_out;
  const int *klist;                           // level of the bz-th slab, or null = identity
  using TS = Tp2dScratch<TI, TJ>;
  using DS = DelnScratch<TI, TJ>;
  static constexpr int nQ = (TI + 6) * (TJ + 6);
  static constexpr int nScr = TS::total > DS::total ? TS::total : DS::total;
  static constexpr int nFXt = (TI + 1) * TJ, nFYt = TI * (TJ + 1);
  static constexpr int lds_doubles = nQ + nScr + nFXt + nFYt + TI * TJ;
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = klist ? klist[bz] : bz;
    const TileBox b = make_box<TI, TJ>(g, bx, by);
    const int i0 = b.i0, j0 = b.j0;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const double *cx = crx + oCX, *xf = xfx + oCX, *cy = cry + oCY, *yf = yfx + oCY;
    double *p = lds;
    const Tile sq{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;
    double *scr = p; p += nScr;
    const Tile sfx{p, i0, j0, TI + 1}; p += nFXt;
    const Tile sfy{p, i0, j0, TI}; p += nFYt;
    const Tile cz{p, i0, j0, TI}; p += TI * TJ;
    load_tile<TI + 6, TJ + 6>(sq, zh + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    FV3_SYNC();
    tp2d_tile<TI, TJ>(g, b, tid, sq, cx, cy, xf, yf, nullptr, nullptr, hord, scr, sfx, sfy);
    FV3_TILE_FOR(TI, TJ, li_, lj_) {
      const int i = i0 + li_, j = j0 + lj_;
      if (i > b.ilast || j > b.jlast) continue;
      const double ar = g.area[g.iA(i, j)];
      const double x0 = xf[g.iCX(i, j)], x1 = xf[g.iCX(i + 1, j)], y0 = yf[g.iCY(i, j)], y1 = yf[g.iCY(i, j + 1)];
      const double fx0 = sfx(i, j) * x0, fx1 = sfx(i + 1, j) * x1, fy0 = sfy(i, j) * y0, fy1 = sfy(i, j + 1) * y1;
      const double rax = ar + x0 - x1, ray = ar + y0 - y1;
      cz(i, j) = (sq(i, j) * ar + fx0 - fx1 + fy0 - fy1) / (rax + ray - ar);
    }
    FV3_SYNC();
    const double dmp = damp[k];
    if (dmp > 1.E-5) {
      Tile fxd, fyd;
      deln_tile<TI, TJ>(g, b, tid, sq, ndif[k], dmp, true, scr, fxd, fyd);
      FV3_TILE_FOR(TI, TJ, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        zh_out[oA + g.iA(i, j)] =
            cz(i, j) + (fxd(i, j) - fxd(i + 1, j) + fyd(i, j) - fyd(i, j + 1)) * g.rarea[g.iA(i, j)];
      }
    } else {
      FV3_TILE_FOR(TI, TJ, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        zh_out[oA + g.iA(i, j)] = cz(i, j);
      }
    }
  }
};

// update_dz_d :303-319.  The heights of kDep interfaces above are in flight while one is fixed (in place: a slot is read before the
// sweep reaches -- and may write -- its interface); an interface the fix leaves as it was (nearly all of them) is not written back.
struct ZhLimit {
  static constexpr int kDep = 8;
  Grid g;
  int km;
  double rdt;
  const double *zs;
  double *zh, *ws;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const int o = g.iA(i, j);
      double nb[kDep];
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++) nb[s] = zh[(size_t)(km - 1 - s > 0 ? km - 1 - s : 0) * nA + o];   // interface km - s (1-based)
      double below = zh[(size_t)km * nA + o];
      ws[g.iCC(i, j)] = (zs[o] - below) * rdt;
      auto fix = [&](int k, double z) {
        const double v = dmax(z, below + kDzMin);
        if (v != z) zh[(size_t)(k - 1) * nA + o] = v;
        below = v;
      };
      int k0 = km;
      for (; k0 - kDep >= 0; k0 -= kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const int k = k0 - s;
          const double z = nb[s];
          nb[s] = zh[(size_t)(k - 1 - kDep > 0 ? k - 1 - kDep : 0) * nA + o];
          fix(k, z);
        }
      }
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++)
        if (k0 - s >= 1) fix(k0 - s, nb[s]);
    }
  }
};

struct ZhFromDelz {  // dyn_core.F90:370-385: gz(npz+1) = zs; gz(k) = gz(k+1) - delz(k)  (compute domain)
  Grid g;
  int km;
  const double *zs, *delz;
  double *zh;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA(), nCC = g.nCC();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const int o = g.iA(i, j), occ = g.iCC(i, j);
      double z = zs[o];
      zh[(size_t)km * nA + o] = z;
      for (int k = km; k >= 1; k--) {
        z = z - delz[(size_t)(k - 1) * nCC + occ];
        zh[(size_t)(k - 1) * nA + o] = z;
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// p_grad_c, dyn_core.F90:1635-1694.  One thread per cell (i, j) of [is, ie+1] x [js, je+1] marching down its column (see NhPGrad
// below): uc between the cells (i - 1, j) and (i, j), vc between (i, j - 1) and (i, j); the interface above in registers, kDep
// interfaces ahead in flight (clamped addresses, no branch around a load).
#ifndef FV3_PGRADC_KDEP
#define FV3_PGRADC_KDEP 4   // 3: 0.272 ms, 4: 0.272, 6: 0.292, 8: 0.291
#endif
#ifndef FV3_PGRAD_KDEP
#define FV3_PGRAD_KDEP 2   // measured at C384 L127 (tools/a2b_ab.py): 2: 0.322 ms, 3: 0.350, 4: 0.344, 5: 0.352 (registers / wavefronts per SIMD)
#endif
struct PGradC {
  static constexpr int kDep = FV3_PGRADC_KDEP;
  Grid g;
  double dt2;
  int hydrostatic;
  const double *delpc, *pkc, *gz;
  double *uc, *vc;
  FV3_HD int ncol() const { return (g.nx + 1) * (g.ny + 1); }
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int km = g.npz, w = g.nx + 1;
    const size_t nA = g.nA(), nU = g.nU(), nV = g.nV();
    const double *FV3_RESTRICT PK = pkc, *FV3_RESTRICT GZ = gz, *FV3_RESTRICT DP = delpc;
    FV3_COL_FOR(c, ncol()) {
      const int i = g.is + c % w, j = g.js + c / w;
      const bool do_u = j <= g.je, do_v = i <= g.ie;
      const int o = g.iA(i, j), ow = g.iA(i - 1, j), os = g.iA(i, j - 1);
      const int ou = g.iV(i, do_u ? j : g.je), ov = g.iU(do_v ? i : g.ie, j);
      double *FV3_RESTRICT pu = uc + ou, *FV3_RESTRICT pv = vc + ov;
      const double rdu = g.rdxc[ou], rdv = g.rdyc[ov];
      struct Lev { double pk, pkw, pks, gz, gzw, gzs, dp, dpw, dps, u, v; };   // interface l + 1 and layer l (0-based)
      auto fetch = [&](int l) {
        const size_t o1 = (size_t)(l + 1 < km ? l + 1 : km) * nA, lc = (size_t)(l < km ? l : km - 1), o0 = lc * nA;
        return Lev{PK[o1 + o], PK[o1 + ow], PK[o1 + os], GZ[o1 + o], GZ[o1 + ow], GZ[o1 + os],
                   hydrostatic ? 0. : DP[o0 + o], hydrostatic ? 0. : DP[o0 + ow], hydrostatic ? 0. : DP[o0 + os], pu[lc * nV], pv[lc * nU]};
      };
      Lev nb[kDep];
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++) nb[s] = fetch(s);
      double pk0 = PK[o], pk0w = PK[ow], pk0s = PK[os], gz0 = GZ[o], gz0w = GZ[ow], gz0s = GZ[os];
      auto layer = [&](int l, const Lev &n) {
        const double wk0 = hydrostatic ? n.pk - pk0 : n.dp;
        const double wkw = hydrostatic ? n.pkw - pk0w : n.dpw, wks = hydrostatic ? n.pks - pk0s : n.dps;
        const double un = n.u + dt2 * rdu / (wkw + wk0) * ((n.gzw - gz0) * (n.pk - pk0w) + (gz0w - n.gz) * (n.pkw - pk0));
        const double vn = n.v + dt2 * rdv / (wks + wk0) * ((n.gzs - gz0) * (n.pk - pk0s) + (gz0s - n.gz) * (n.pks - pk0));
        if (do_u) pu[(size_t)l * nV] = un;
        if (do_v) pv[(size_t)l * nU] = vn;
        pk0 = n.pk; pk0w = n.pkw; pk0s = n.pks; gz0 = n.gz; gz0w = n.gzw; gz0s = n.gzs;
      };
      int l0 = 0;
      for (; l0 + kDep <= km; l0 += kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const Lev n = nb[s];
          nb[s] = fetch(l0 + s + kDep);
          layer(l0 + s, n);
        }
      }
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++)
        if (l0 + s < km) layer(l0 + s, nb[s]);
    }
  }
};

// The corners (i, j) and (i, j + 1) of a2b_ord4 away from the face edges, from a tile of cells in LDS: their 4 x 5 cells once into
// registers (the 4 x 4 blocks of neighbouring corners share 12 cells, and qx / qy of one corner read the same 16), then the sums of the
// reference.  sum_form 0: the combined sum of the grid_type >= 3 branch (a2b_edge.F90:292-315); 1: qout = 0.5 (qxx + qyy) with the two
// 4-point sums formed separately (the cubed-sphere branch, :236-286).
FV3_HD void a2b_corner_pair(const Tile &s, int i, int j, int sum_form, double (&qo)[2]) {
  constexpr double a1 = 0.5625, a2 = -0.0625, b1 = 7. / 12., b2 = -1. / 12.;
  constexpr int NV = 2;
  double cl[NV + 3][4];
  for (int r = 0; r < NV + 3; r++)
    for (int t = 0; t < 4; t++) cl[r][t] = s(i - 2 + t, j - 2 + r);
  double qxr[NV + 3];
  for (int r = 0; r < NV + 3; r++) qxr[r] = b1 * (cl[r][1] + cl[r][2]) + b2 * (cl[r][0] + cl[r][3]);
  for (int d = 0; d < NV; d++) {
    double qx[4], qy[4];
    for (int t = 0; t < 4; t++) {
      qx[t] = qxr[d + t];
      qy[t] = b1 * (cl[d + 1][t] + cl[d + 2][t]) + b2 * (cl[d][t] + cl[d + 3][t]);
    }
    if (sum_form)
      qo[d] = 0.5 * ((a2 * (qx[0] + qx[3]) + a1 * (qx[1] + qx[2])) + (a2 * (qy[0] + qy[3]) + a1 * (qy[1] + qy[2])));
    else
      qo[d] = 0.5 * (a1 * (qx[1] + qx[2] + qy[1] + qy[2]) + a2 * (qx[0] + qx[3] + qy[0] + qy[3]));
  }
}

// a2b_ord4, doubly periodic branch (a2b_edge.F90:292-315), for up to 4 fields of one level: corner values on
// [is, ie+1] x [js, je+1] written to separate B-position slabs stored with the A layout (corner (i,j) at iA(i,j)).
template <int TI, int TJ>
struct A2BCorners {
  Grid g;
  const double *in[4];
  double *out[4];
  int nlev[4];      // number of levels of each field (npz+1 or npz)
  int nf;
  double scale[4];        // value = in * scale at load time (gz = zh*grav, dyn_core.F90:982-989); 1.0 = none
  double top[4];          // level-1 overrides (nh_p_grad :1732-1738, one_grad_p :1950-1955) ...
  int override_mask;      // ... of the fields whose bit is set
  int sum_form = 0;       // 1: qout = 0.5*(qxx + qyy) with the two 4-point sums formed separately (the cubed-sphere branch,
                          // a2b_edge.F90:236-286) instead of the combined sum of the grid_type >= 3 branch (:292-315)
  static constexpr int W = TI + 5, H = TJ + 5;  // corners [i0, i0+TI] need cells [i0-2, i0+TI+1]
  // A tile per field: the cells of every field of the level are requested together and go to LDS behind ONE barrier (that orders LDS
  // only: inputs and outputs are distinct arrays), so no field waits for the stores of the one before it.  The first form pushed the
  // fields through one tile, __syncthreads() (= s_waitcnt vmcnt(0): loads AND stores) on either side of every one of them.
  static constexpr int lds_doubles = 4 * W * H;
  static constexpr int kIt = (W * H + kNT - 1) / kNT;
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = bz;
    const int i0 = g.is + bx * TI, j0 = g.js + by * TJ;
    bool act[4], ovr[4];
    for (int f = 0; f < 4; f++) {
      ovr[f] = f < nf && k < nlev[f] && k == 0 && ((override_mask >> f) & 1);
      act[f] = f < nf && k < nlev[f] && !ovr[f];
    }
    double v[4][kIt];
    FV3_UNROLL_ALL
    for (int f = 0; f < 4; f++) {
      if (!act[f]) continue;
      const double *src = in[f] + (size_t)k * g.nA();
      FV3_UNROLL_ALL
      for (int it = 0; it < kIt; it++) {
        const int idx = tid + it * kNT, li = idx % W, lj = idx / W;
        const int i = i0 - 2 + li, j = j0 - 2 + lj;
        v[f][it] = 0.;
        if (idx < W * H && i >= g.isd && i <= g.ied && j >= g.jsd && j <= g.jed) v[f][it] = src[(j - g.jsd) * g.nid + (i - g.isd)];
      }
    }
    FV3_UNROLL_ALL
    for (int f = 0; f < 4; f++) {
      if (!act[f]) continue;
      double *t = lds + f * (W * H);
      FV3_UNROLL_ALL
      for (int it = 0; it < kIt; it++) {
        const int idx = tid + it * kNT;
        if (idx < W * H) t[idx] = scale[f] != 1.0 ? v[f][it] * scale[f] : v[f][it];
      }
    }
    FV3_SYNC_LDS();
    for (int f = 0; f < 4; f++) {
      if (!act[f] && !ovr[f]) continue;
      double *o = out[f] + (size_t)k * g.nA();
      if (ovr[f]) {
        FV3_TILE_FOR(TI, TJ, li_, lj_) {
          const int i = i0 + li_, j = j0 + lj_;
          if (i > g.ie + 1 || j > g.je + 1) continue;
          o[g.iA(i, j)] = top[f];
        }
        continue;
      }
      const Tile s{lds + f * (W * H), i0 - 2, j0 - 2, W};
      // a thread takes the NV corners (i, j .. j + NV - 1): their 4 x (NV + 3) cells once into registers (the 4 x 4 blocks of
      // neighbouring corners share 12 cells, and qx / qy of one corner read the same 16), then the sums of the reference
      constexpr int NV = 2;   // 4 (4 x 7 cells, 32 x 32 tiles) was measured slower: 0.43 against 0.40 ms
      static_assert(TJ % 2 == 0, "A2BCorners: TJ must be even");
      for (int idx = tid; idx < TI * (TJ / NV); idx += kNT) {
        const int i = i0 + idx % TI, j = j0 + NV * (idx / TI);
        if (i > g.ie + 1 || j > g.je + 1) continue;
        double qo[NV];
        a2b_corner_pair(s, i, j, sum_form, qo);
        for (int d = 0; d < NV; d++)
          if (j + d <= g.je + 1) o[g.iA(i, j + d)] = qo[d];
      }
    }
  }
};

// nh_p_grad :1746-1790 on precomputed corner values; SPLIT: split_p_grad :1795-1900 (beta > 0) -- the hydrostatic part of the
// gradient of the previous substep (du / dv: U / V x npz, zero before the first call: dyn_core.F90:278-283) enters with weight beta,
// the current one with 1 - beta, and is stored for the next substep.
// One thread per corner (i, j) of [is, ie+1] x [js, je+1], marching down its column: u between the corners (i, j) and (i + 1, j), v
// between (i, j) and (i, j + 1); the corner values of the interface above stay in registers, those of kDep interfaces ahead are in
// flight (a rolling buffer refilled as it is used, clamped addresses).  The first form of this kernel was a launch per level: both
// interfaces of a layer loaded by every thread, 27 loads per cell for 14.
template <bool SPLIT>
struct NhPGrad {
  static constexpr int kDep = FV3_PGRAD_KDEP;
  Grid g;
  double dt;
  const double *pp, *pk, *gz, *dpc;  // corner slabs: pp, pk, gz (npz+1 levels), delp (npz levels)
  double *u, *v;
  double beta = 0.;
  double *du = nullptr, *dv = nullptr;
  FV3_HD int ncol() const { return (g.nx + 1) * (g.ny + 1); }
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int km = g.npz, w = g.nx + 1;
    const size_t nA = g.nA(), nU = g.nU(), nV = g.nV();
    const double *FV3_RESTRICT PP = pp, *FV3_RESTRICT PK = pk, *FV3_RESTRICT GZ = gz, *FV3_RESTRICT W1 = dpc;
    FV3_COL_FOR(c, ncol()) {
      const int i = g.is + c % w, j = g.js + c / w;
      const bool do_u = i <= g.ie, do_v = j <= g.je;
      const int o = g.iA(i, j), oe = g.iA(i + 1, j), on = g.iA(i, j + 1);
      const int ou = g.iU(do_u ? i : g.ie, j), ov = g.iV(i, do_v ? j : g.je);
      double *FV3_RESTRICT pu = u + ou, *FV3_RESTRICT pv = v + ov;
      double *FV3_RESTRICT qu = SPLIT ? du + ou : nullptr, *FV3_RESTRICT qv = SPLIT ? dv + ov : nullptr;
      const double rdu = g.rdx[ou], rdv = g.rdy[ov];
      struct Lev { double pp, ppe, ppn, pk, pke, pkn, gz, gze, gzn, w, we, wn, u, v, du, dv; };   // interface l + 1, layer l (0-based)
      auto fetch = [&](int l) {
        const size_t o1 = (size_t)(l + 1 < km ? l + 1 : km) * nA, lc = (size_t)(l < km ? l : km - 1), o0 = lc * nA;
        return Lev{PP[o1 + o], PP[o1 + oe], PP[o1 + on], PK[o1 + o], PK[o1 + oe], PK[o1 + on], GZ[o1 + o], GZ[o1 + oe], GZ[o1 + on],
                   W1[o0 + o], W1[o0 + oe], W1[o0 + on], pu[lc * nU], pv[lc * nV], SPLIT ? qu[lc * nU] : 0., SPLIT ? qv[lc * nV] : 0.};
      };
      Lev nb[kDep];
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++) nb[s] = fetch(s);
      double pp0 = PP[o], pp0e = PP[oe], pp0n = PP[on], pk0 = PK[o], pk0e = PK[oe], pk0n = PK[on], gz0 = GZ[o], gz0e = GZ[oe], gz0n = GZ[on];
      auto layer = [&](int l, const Lev &n) {
        const double wk0 = n.pk - pk0, wke = n.pke - pk0e, wkn = n.pkn - pk0n;
        const double du1 = dt / (wk0 + wke) * ((n.gz - gz0e) * (n.pke - pk0) + (gz0 - n.gze) * (n.pk - pk0e));
        const double du2 = dt / (n.w + n.we) * ((n.gz - gz0e) * (n.ppe - pp0) + (gz0 - n.gze) * (n.pp - pp0e));
        const double dv1 = dt / (wk0 + wkn) * ((n.gz - gz0n) * (n.pkn - pk0) + (gz0 - n.gzn) * (n.pk - pk0n));
        const double dv2 = dt / (n.w + n.wn) * ((n.gz - gz0n) * (n.ppn - pp0) + (gz0 - n.gzn) * (n.pp - pp0n));
        double un, vn;
        if (SPLIT) {
          un = (n.u + beta * n.du + (1. - beta) * du1 + du2) * rdu;                 // :1866
          vn = (n.v + beta * n.dv + (1. - beta) * dv1 + dv2) * rdv;                 // :1885
        } else {
          un = (n.u + du1 + du2) * rdu;
          vn = (n.v + dv1 + dv2) * rdv;
        }
        if (do_u) {
          pu[(size_t)l * nU] = un;
          if (SPLIT) qu[(size_t)l * nU] = du1;
        }
        if (do_v) {
          pv[(size_t)l * nV] = vn;
          if (SPLIT) qv[(size_t)l * nV] = dv1;
        }
        pp0 = n.pp; pp0e = n.ppe; pp0n = n.ppn; pk0 = n.pk; pk0e = n.pke; pk0n = n.pkn; gz0 = n.gz; gz0e = n.gze; gz0n = n.gzn;
      };
      int l0 = 0;
      for (; l0 + kDep <= km; l0 += kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const Lev n = nb[s];
          nb[s] = fetch(l0 + s + kDep);
          layer(l0 + s, n);
        }
      }
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
      for (int s = 0; s < kDep; s++)
        if (l0 + s < km) layer(l0 + s, nb[s]);
    }
  }
};

// A store of one value per thread without a branch around it (a branch around a global store costs the loop an s_waitcnt vmcnt(0) at its
// top, spmd.h "branch-free rows"): through a buffer resource over the level's slab (`base`: wave-uniform), a thread that must not store
// carries an offset beyond num_records and the hardware drops its store.
#ifdef FV3_HOST_EMU
inline void store_if(double *base, size_t, int idx, double x, bool valid) {
  if (valid) base[idx] = x;
}
#else
typedef unsigned fv3_nh_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_if(double *base, size_t nelem, int idx, double x, bool valid) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(fv3_nh_u2, x), __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(nelem * 8), 0x00020000),
                                        valid ? idx * 8 : (int)0x80000000u, 0, 0);
}
#endif

// nh_p_grad (dyn_core.F90:1697-1792) in ONE kernel on a domain without face edges (grid_type >= 3): a2b_ord4 of pp, pk, gz and delp and the
// gradient that reads the corner values, which then never go to memory (A2BCorners + NhPGrad: 64 B per cell written and read back beside the
// 64 the routine needs).  A workgroup owns a TI x TJ tile of corners and KC layers: interface after interface it stages the cells of the
// four fields in LDS (the loads of the next interface are in flight while this one is worked on), forms the (TI + 1) x (TJ + 1) corner
// values a thread's own corner and its east / north neighbours need into a second set of tiles, and updates u, v of the layer above from
// them and the corner values of the interface above, which each thread kept in registers.  The arithmetic is that of the two kernels
// (a2b_corner_pair, the statements of NhPGrad::layer): the same bits.
#ifndef PGF_UVPRE
#define PGF_UVPRE 0
#endif
template <int TI, int TJ>
struct NhPGradFused {
  static constexpr int KC = 16;
  static constexpr int W = TI + 5, H = TJ + 5, CW = TI + 1, CH = TJ + 1;
  static constexpr int lds_doubles = 4 * W * H + 4 * CW * CH;
  static constexpr int kIt = (W * H + kNT - 1) / kNT;        // cells a thread stages per field
  static constexpr int kPt = (TI * TJ + kNT - 1) / kNT;      // corners whose u, v a thread updates
  static_assert(CH % 2 == 1, "NhPGradFused: the corner rows are taken in pairs and one single row");
  Grid g;
  double dt, gz_scale, top_value;
  const double *pp, *pk, *gz, *delp;   // A x (npz+1), A x (npz+1), A x (npz+1), A x npz
  double *u, *v;
  FV3_HD int nchunks() const { return (g.npz + KC - 1) / KC; }
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int km = g.npz, k0 = bz * KC, k1 = (k0 + KC < km) ? k0 + KC : km;
    const int i0 = g.is + bx * TI, j0 = g.js + by * TJ;
    const size_t nA = g.nA(), nU = g.nU(), nV = g.nV();
    double *cells = lds, *corn = lds + 4 * W * H;
    double vq[4][kIt];
    // the cells of interface l (pp, pk, gz) and of the layer above it (delp of layer l - 1; nothing at the chunk's first interface).  Every
    // load unconditional, from an address inside the array (a load under a branch is waited for at the join); cells outside the array
    // belong to corners outside [is, ie + 1] x [js, je + 1], whose values are dropped
    int coff[kIt];
    for (int it = 0; it < kIt; it++) {
      const int e0 = tid + it * kNT, idx = e0 < W * H ? e0 : W * H - 1, li = idx % W, lj = idx / W;
      int i = i0 - 2 + li, j = j0 - 2 + lj;
      i = i < g.isd ? g.isd : (i > g.ied ? g.ied : i);
      j = j < g.jsd ? g.jsd : (j > g.jed ? g.jed : j);
      coff[it] = (j - g.jsd) * g.nid + (i - g.isd);
    }
    auto issue = [&](int l) {
      const double *src[4] = {pp + (size_t)l * nA, pk + (size_t)l * nA, gz + (size_t)l * nA, delp + (size_t)(l > 0 ? l - 1 : 0) * nA};
      FV3_UNROLL_ALL
      for (int f = 0; f < 4; f++) {
        FV3_UNROLL_ALL
        for (int it = 0; it < kIt; it++) vq[f][it] = src[f][coff[it]];
      }
    };
    double pv[kPt][9];   // pp, pk, gz of the interface above at the thread's corner, its east and its north neighbour
    double rdu[kPt], rdv[kPt];
    for (int p = 0; p < kPt; p++) {
      const int idx = tid + p * kNT, i = i0 + idx % TI, j = j0 + idx / TI;
      const bool in = idx < TI * TJ && i <= g.ie + 1 && j <= g.je + 1;
      rdu[p] = (in && i <= g.ie) ? g.rdx[g.iU(i, j)] : 0.;
      rdv[p] = (in && j <= g.je) ? g.rdy[g.iV(i, j)] : 0.;
      for (int q = 0; q < 9; q++) pv[p][q] = 0.;
    }
    issue(k0);
    for (int l = k0; l <= k1; l++) {
      FV3_UNROLL_ALL
      for (int f = 0; f < 4; f++) {
        double *t = cells + f * (W * H);
        FV3_UNROLL_ALL
        for (int it = 0; it < kIt; it++) {
          const int idx = tid + it * kNT;
          if (idx < W * H) t[idx] = (f == 2 && gz_scale != 1.0) ? vq[f][it] * gz_scale : vq[f][it];
        }
      }
      FV3_SYNC_LDS();
      if (l < k1) issue(l + 1);
#if PGF_UVPRE
      // u, v of the layer above: requested now, used behind the corner values
      double uo[kPt], vo[kPt];
      for (int p = 0; p < kPt; p++) {
        const int idx = tid + p * kNT, i = i0 + idx % TI, j = j0 + idx / TI;
        const int iu = i < g.ie ? i : g.ie, ju = j < g.je + 1 ? j : g.je + 1, iv = i < g.ie + 1 ? i : g.ie + 1, jv = j < g.je ? j : g.je;
        const size_t lc = (size_t)(l > 0 ? l - 1 : 0);
        uo[p] = u[lc * nU + g.iU(iu, ju)];
        vo[p] = v[lc * nV + g.iV(iv, jv)];
      }
#endif
      // corner values of this interface (and of the layer above it) on [i0, i0 + TI] x [j0, j0 + TJ]
      for (int f = 0; f < 4; f++) {
        if (f == 3 && l == k0) continue;
        double *c = corn + f * (CW * CH);
        if (l == 0 && f < 2) {                      // nh_p_grad :1732-1738: the top interface
          for (int idx = tid; idx < CW * CH; idx += kNT) c[idx] = f == 0 ? 0. : top_value;
          continue;
        }
        const Tile s{cells + f * (W * H), i0 - 2, j0 - 2, W};
        for (int idx = tid; idx < CW * ((CH + 1) / 2); idx += kNT) {
          const int ci = idx % CW, cj = 2 * (idx / CW);
          double qo[2];
          a2b_corner_pair(s, i0 + ci, j0 + (cj + 1 < CH ? cj : cj - 1), 0, qo);   // the single last row: as the second of its pair
          if (cj + 1 < CH) {
            c[cj * CW + ci] = qo[0];
            c[(cj + 1) * CW + ci] = qo[1];
          } else {
            c[cj * CW + ci] = qo[1];
          }
        }
      }
      FV3_SYNC_LDS();
      const double *cpp = corn, *cpk = corn + CW * CH, *cgz = corn + 2 * CW * CH, *cw1 = corn + 3 * CW * CH;
      for (int p = 0; p < kPt; p++) {
        const int idx = tid + p * kNT;
        if (idx >= TI * TJ) continue;
        const int ci = idx % TI, cj = idx / TI, i = i0 + ci, j = j0 + cj;
        const int o = cj * CW + ci, oe = o + 1, on = o + CW;
        const double n_pp = cpp[o], n_ppe = cpp[oe], n_ppn = cpp[on], n_pk = cpk[o], n_pke = cpk[oe], n_pkn = cpk[on];
        const double n_gz = cgz[o], n_gze = cgz[oe], n_gzn = cgz[on];
        {
          const double pp0 = pv[p][0], pp0e = pv[p][1], pp0n = pv[p][2], pk0 = pv[p][3], pk0e = pv[p][4], pk0n = pv[p][5];
          const double gz0 = pv[p][6], gz0e = pv[p][7], gz0n = pv[p][8];
          const double n_w = cw1[o], n_we = cw1[oe], n_wn = cw1[on];
          const size_t lc = (size_t)(l > 0 ? l - 1 : 0);
          const bool in = l > k0 && i <= g.ie + 1 && j <= g.je + 1;
#if PGF_UVPRE
          const double u0 = uo[p], v0 = vo[p];
#else
          const int iu = i < g.ie ? i : g.ie, ju = j < g.je + 1 ? j : g.je + 1, iv = i < g.ie + 1 ? i : g.ie + 1, jv = j < g.je ? j : g.je;
          const double u0 = u[lc * nU + g.iU(iu, ju)], v0 = v[lc * nV + g.iV(iv, jv)];
#endif
          const double wk0 = n_pk - pk0, wke = n_pke - pk0e, wkn = n_pkn - pk0n;
          const double du1 = dt / (wk0 + wke) * ((n_gz - gz0e) * (n_pke - pk0) + (gz0 - n_gze) * (n_pk - pk0e));
          const double du2 = dt / (n_w + n_we) * ((n_gz - gz0e) * (n_ppe - pp0) + (gz0 - n_gze) * (n_pp - pp0e));
          store_if(u + lc * nU, nU, g.iU(i, j), (u0 + du1 + du2) * rdu[p], in && i <= g.ie);
          const double dv1 = dt / (wk0 + wkn) * ((n_gz - gz0n) * (n_pkn - pk0) + (gz0 - n_gzn) * (n_pk - pk0n));
          const double dv2 = dt / (n_w + n_wn) * ((n_gz - gz0n) * (n_ppn - pp0) + (gz0 - n_gzn) * (n_pp - pp0n));
          store_if(v + lc * nV, nV, g.iV(i, j), (v0 + dv1 + dv2) * rdv[p], in && j <= g.je);
        }
        pv[p][0] = n_pp; pv[p][1] = n_ppe; pv[p][2] = n_ppn; pv[p][3] = n_pk; pv[p][4] = n_pke; pv[p][5] = n_pkn;
        pv[p][6] = n_gz; pv[p][7] = n_gze; pv[p][8] = n_gzn;
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// pk3_halo / pln_halo: the 2-wide ring [is-2,ie+2]^2 minus the compute domain (dyn_core.F90:1395-1496)
// The ring is 4 (nx + ny + 4) columns: they are enumerated compactly (a launch over the whole (nx + 4) x (ny + 4) box left two lanes
// of most wavefronts working), and a workgroup takes NC of them in two phases: NC threads form the hydrostatic pressures of a
// column each, in the reference's order, into LDS; then all threads take the logarithm / the power of one (column, level) each.
struct Pk3Halo {
  static constexpr int NC = 8;
  Grid g;
  int npz, use_logp;
  double ptop, akap;
  const double *delp;
  double *pk3;
  static size_t lds_doubles(int npz) { return (size_t)NC * npz; }
  FV3_HD int ring() const { return 4 * (g.nx + 4) + 4 * g.ny; }
  FV3_HD int ring_column(int r) const {   // r-th column of the ring: the two rows below, the two rows above, then the sides row by row
    const int w = g.nx + 4;
    int i, j;
    if (r < 2 * w) {
      j = g.js - 2 + r / w; i = g.is - 2 + r % w;
    } else if (r < 4 * w) {
      r -= 2 * w;
      j = g.je + 1 + r / w; i = g.is - 2 + r % w;
    } else {
      r -= 4 * w;
      j = g.js + r / 4;
      const int s = r % 4;
      i = s < 2 ? g.is - 2 + s : g.ie - 1 + s;
    }
    return g.iA(i, j);
  }
  FV3_HD void operator()(int bx, int, int, int tid, double *lds) const {
    const size_t nA = g.nA();
    const int nr = ring();
    for (int t = tid; t < NC; t += kNT) {
      const int r = bx * NC + t;
      if (r >= nr) continue;
      const int o = ring_column(r);
      double pet = ptop;
      for (int k = 1; k <= npz; k++) {
        pet = pet + delp[(size_t)(k - 1) * nA + o];
        lds[t * npz + k - 1] = pet;
      }
    }
    FV3_SYNC();
    for (int idx = tid; idx < NC * npz; idx += kNT) {
      const int t = idx % NC, k = idx / NC + 1, r = bx * NC + t;
      if (r >= nr) continue;
      const double pet = lds[t * npz + k - 1];
      pk3[(size_t)k * nA + ring_column(r)] = use_logp ? dlog(pet) : dexp(akap * dlog(pet));
    }
  }
};

struct PeHalo {  // pe_halo, dyn_core.F90:1498-1526: the 1-wide ring of pe(is-1:ie+1, npz+1, js-1:je+1)
  Grid g;
  int npz;
  double ptop;
  const double *delp;
  double *pe;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int w = g.nx + 2, ncol = w * (g.ny + 2);
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is - 1 + c % w, j = g.js - 1 + c / w;
      if (i >= g.is && i <= g.ie && j >= g.js && j <= g.je) continue;
      // the reference fills (is-1|ie+1, js:je) and (is-1:ie+1, js-1|je+1): all ring points
      const int o = g.iA(i, j);
      const size_t pb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (npz + 1) + (i - (g.is - 1));
      double p = ptop;
      pe[pb] = p;
      for (int k = 1; k <= npz; k++) {
        p = p + delp[(size_t)(k - 1) * nA + o];
        pe[pb + (size_t)k * (g.nx + 2)] = p;
      }
    }
  }
};

struct Geopk {  // geopk, dyn_core.F90:2202-2353 (use_cond = .false.)
  Grid g;
  int km, CG;
  double ptop, akap, cp_air, ptk;
  const double *delp, *hs, *pt;
  double *pe, *peln, *pk, *gz, *pkz;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int e = CG ? 1 : 2;
    const int w = g.nx + 2 * e, ncol = w * (g.ny + 2 * e);
    const size_t nA = g.nA(), nCC = g.nCC();
    const double peln1 = dlog(ptop);
    FV3_COL_FOR(c, ncol) {
      const int i = g.is - e + c % w, j = g.js - e + c / w;
      const int o = g.iA(i, j);
      const bool in_pe = (j > g.js - 2 && j < g.je + 2 && i >= g.is - 1 && i <= g.ie + 1);
      const bool in_c = (j >= g.js && j <= g.je && i >= g.is && i <= g.ie);
      const size_t pb = in_pe ? (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1)) : 0;
      const size_t lb = in_c ? (size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is) : 0;
      double p1d = ptop;
      pk[o] = ptk;
      if (in_c) peln[lb] = peln1;
      if (in_pe) pe[pb] = ptop;
      for (int k = 2; k <= km + 1; k++) {
        p1d = p1d + delp[(size_t)(k - 2) * nA + o];
        const double logp = dlog(p1d);
        pk[(size_t)(k - 1) * nA + o] = dexp(akap * logp);
        if (in_pe) pe[pb + (size_t)(k - 1) * (g.nx + 2)] = p1d;
        if (in_c) peln[lb + (size_t)(k - 1) * g.nx] = logp;
      }
      double zb = hs[o];
      gz[(size_t)km * nA + o] = zb;
      for (int k = km; k >= 1; k--) {
        zb = zb + cp_air * pt[(size_t)(k - 1) * nA + o] * (pk[(size_t)k * nA + o] - pk[(size_t)(k - 1) * nA + o]);
        gz[(size_t)(k - 1) * nA + o] = zb;
      }
      if (!CG && in_c) {
        for (int k = 1; k <= km; k++)
          pkz[(size_t)(k - 1) * nCC + g.iCC(i, j)] =
              (pk[(size_t)k * nA + o] - pk[(size_t)(k - 1) * nA + o]) /
              (akap * (peln[lb + (size_t)k * g.nx] - peln[lb + (size_t)(k - 1) * g.nx]));
      }
    }
  }
};

// The same in phases over LDS, for faces too small to fill the chip with one thread per column (a C96 face has 157 wavefronts for
// 1 024 SIMDs, each running 79 logarithms and powers one after the other: 130 us per call, a third of the stream time of BASELINE
// config 2).  A workgroup takes NC consecutive columns: (a) NC threads sum the pressures of a column each, in the reference's order,
// into LDS; (b) all threads take the logarithm and the power of one (column, interface) each and write pe, peln, pk; (c) NC threads
// run the hydrostatic integral upwards with pk from LDS; (d) all threads form pkz.  The values are those of the kernel above bit for
// bit (the same operations on the same operands).  The loads of the two serial phases go through rolling buffers of kDep levels.
struct GeopkPhased {
  static constexpr int NC = 16, kDep = 8;
  Grid g;
  int km, CG;
  double ptop, akap, cp_air, ptk;
  const double *delp, *hs, *pt;
  double *pe, *peln, *pk, *gz, *pkz;
  static size_t lds_doubles(int km) { return (size_t)2 * NC * (km + 1); }
  FV3_HD int ncol() const { const int e = CG ? 1 : 2; return (g.nx + 2 * e) * (g.ny + 2 * e); }
  FV3_HD void operator()(int bx, int, int, int tid, double *lds) const {
    const int e = CG ? 1 : 2, nk = km + 1;
    const int w = g.nx + 2 * e, nc = ncol();
    const size_t nA = g.nA(), nCC = g.nCC();
    double *P = lds, *K = lds + (size_t)NC * nk;   // [column][interface]: p, then log p; pk
    const double *FV3_RESTRICT DP = delp, *FV3_RESTRICT PT = pt;
    double *FV3_RESTRICT GZ = gz;
    struct Col { int o; bool in_pe, in_c; size_t pb, lb; int icc; bool ok; };
    auto column = [&](int t) {
      const int c = bx * NC + t;
      Col q;
      q.ok = c < nc;
      const int cc = q.ok ? c : nc - 1;
      const int i = g.is - e + cc % w, j = g.js - e + cc / w;
      q.o = g.iA(i, j);
      q.in_pe = (j > g.js - 2 && j < g.je + 2 && i >= g.is - 1 && i <= g.ie + 1);
      q.in_c = (j >= g.js && j <= g.je && i >= g.is && i <= g.ie);
      q.pb = q.in_pe ? (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1)) : 0;
      q.lb = q.in_c ? (size_t)(j - g.js) * g.nx * (km + 1) + (i - g.is) : 0;
      q.icc = q.in_c ? g.iCC(i, j) : 0;
      return q;
    };
    // (a) p(k) = ptop + sum delp
    for (int t = tid; t < NC; t += kNT) {
      const Col q = column(t);
      double nb[kDep];
      for (int s = 0; s < kDep; s++) nb[s] = DP[(size_t)(s < km ? s : km - 1) * nA + q.o];
      double p1d = ptop;
      P[t * nk] = p1d;
      int l0 = 0;
      for (; l0 + kDep <= km; l0 += kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const double d = nb[s];
          const int ln = l0 + s + kDep;
          nb[s] = DP[(size_t)(ln < km ? ln : km - 1) * nA + q.o];
          p1d = p1d + d;
          P[t * nk + l0 + s + 1] = p1d;
        }
      }
      for (int s = 0; s < kDep; s++)
        if (l0 + s < km) {
          p1d = p1d + nb[s];
          P[t * nk + l0 + s + 1] = p1d;
        }
    }
    FV3_SYNC();
    // (b) log p, p^kappa of every (column, interface)
    const double peln1 = dlog(ptop);
    for (int idx = tid; idx < NC * nk; idx += kNT) {
      const int t = idx % NC, k = idx / NC;   // interface k + 1
      const Col q = column(t);
      const double p1d = P[t * nk + k];
      const double logp = k == 0 ? peln1 : dlog(p1d);
      const double pkv = k == 0 ? ptk : dexp(akap * logp);
      P[t * nk + k] = logp;
      K[t * nk + k] = pkv;
      if (!q.ok) continue;
      pk[(size_t)k * nA + q.o] = pkv;
      if (q.in_pe) pe[q.pb + (size_t)k * (g.nx + 2)] = p1d;
      if (q.in_c) peln[q.lb + (size_t)k * g.nx] = logp;
    }
    FV3_SYNC();
    // (c) gz(k) = gz(k+1) + cp pt(k) (pk(k+1) - pk(k)), from the surface up
    for (int t = tid; t < NC; t += kNT) {
      const Col q = column(t);
      if (!q.ok) continue;
      double nb[kDep];
      for (int s = 0; s < kDep; s++) nb[s] = PT[(size_t)(km - 1 - s > 0 ? km - 1 - s : 0) * nA + q.o];
      double zb = hs[q.o];
      GZ[(size_t)km * nA + q.o] = zb;
      int k0 = km;
      for (; k0 - kDep >= 0; k0 -= kDep) {
#ifndef FV3_HOST_EMU
#pragma unroll
#endif
        for (int s = 0; s < kDep; s++) {
          const int k = k0 - s;               // layer k (1-based)
          const double ptv = nb[s];
          nb[s] = PT[(size_t)(k - 1 - kDep > 0 ? k - 1 - kDep : 0) * nA + q.o];
          zb = zb + cp_air * ptv * (K[t * nk + k] - K[t * nk + k - 1]);
          GZ[(size_t)(k - 1) * nA + q.o] = zb;
        }
      }
      for (int s = 0; s < kDep; s++)
        if (k0 - s >= 1) {
          const int k = k0 - s;
          zb = zb + cp_air * nb[s] * (K[t * nk + k] - K[t * nk + k - 1]);
          GZ[(size_t)(k - 1) * nA + q.o] = zb;
        }
    }
    // (d) pkz (reads LDS only: no barrier needed after (c))
    if (!CG) {
      for (int idx = tid; idx < NC * km; idx += kNT) {
        const int t = idx % NC, k = idx / NC + 1;
        const Col q = column(t);
        if (!q.ok || !q.in_c) continue;
        pkz[(size_t)(k - 1) * nCC + q.icc] = (K[t * nk + k] - K[t * nk + k - 1]) / (akap * (P[t * nk + k] - P[t * nk + k - 1]));
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// dissipative heating after the substep loop (dyn_core.F90:798-803, :1300-1355, del2_cubed :2356-2465)
struct HeatAccum {
  Grid g;
  double *hs3;        // A x npz
  const double *hs2;  // CC x npz
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int n = g.nx * g.ny;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      double *p = hs3 + (size_t)bz * g.nA() + g.iA(i, j);
      *p = *p + hs2[(size_t)bz * g.nCC() + idx];
    }
  }
};

struct Del2Pass {  // one pass of del2_cubed on the box [is-nt, ie+nt] x [js-nt, je+nt], out of place
  Grid g;
  const double *qi;
  double *qo;
  double cd;
  int nt;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int n = g.nid * g.njd;
    const double *q = qi + (size_t)bz * g.nA();
    double *o = qo + (size_t)bz * g.nA();
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.isd + idx % g.nid, j = g.jsd + idx / g.nid;
      double v = q[idx];
      if (g.grid_type < 3) {
        // a cubed-sphere face: the three cells around every cube corner share their mean first (dyn_core.F90:2409-2428; R),
        // copy_corners before the x / y differences when nt > 0 (:2430, :2443) as an index map on the reads (RD); the centre
        // of a corner-region cell is what the last copy (direction 2) left there
        const int npx = g.npx, npy = g.npy, ie = npx - 1, je = npy - 1;
        constexpr double r3 = 1. / 3.;
        auto Q = [&](int ii, int jj) { return q[g.iA(ii, jj)]; };
        auto R = [&](int ii, int jj) -> double {
          if ((ii == 1 && jj == 1) || (ii == 0 && jj == 1) || (ii == 1 && jj == 0)) return (Q(1, 1) + Q(0, 1) + Q(1, 0)) * r3;
          if ((ii == ie && jj == 1) || (ii == npx && jj == 1) || (ii == ie && jj == 0)) return (Q(ie, 1) + Q(npx, 1) + Q(ie, 0)) * r3;
          if ((ii == ie && jj == je) || (ii == npx && jj == je) || (ii == ie && jj == npy)) return (Q(ie, je) + Q(npx, je) + Q(ie, npy)) * r3;
          if ((ii == 1 && jj == je) || (ii == 0 && jj == je) || (ii == 1 && jj == npy)) return (Q(1, je) + Q(0, je) + Q(1, npy)) * r3;
          return Q(ii, jj);
        };
        auto RD = [&](int dir, int ii, int jj) {
          if (nt > 0) copyc_src(dir, npx, npy, ii, jj);
          return R(ii, jj);
        };
        v = R(i, j);
        if (i >= g.is - nt && i <= g.ie + nt && j >= g.js - nt && j <= g.je + nt) {
          const double cx = RD(1, i, j), cy = RD(2, i, j);
          const double fx0 = g.del6_v[g.iV(i, j)] * (RD(1, i - 1, j) - cx);
          const double fx1 = g.del6_v[g.iV(i + 1, j)] * (cx - RD(1, i + 1, j));
          const double fy0 = g.del6_u[g.iU(i, j)] * (RD(2, i, j - 1) - cy);
          const double fy1 = g.del6_u[g.iU(i, j + 1)] * (cy - RD(2, i, j + 1));
          v = cy + cd * g.rarea[idx] * (fx0 - fx1 + fy0 - fy1);
        }
        o[idx] = v;
        continue;
      }
      if (i >= g.is - nt && i <= g.ie + nt && j >= g.js - nt && j <= g.je + nt) {
        const double fx0 = g.del6_v[g.iV(i, j)] * (q[g.iA(i - 1, j)] - v);
        const double fx1 = g.del6_v[g.iV(i + 1, j)] * (v - q[g.iA(i + 1, j)]);
        const double fy0 = g.del6_u[g.iU(i, j)] * (q[g.iA(i, j - 1)] - v);
        const double fy1 = g.del6_u[g.iU(i, j + 1)] * (v - q[g.iA(i, j + 1)]);
        v = v + cd * g.rarea[idx] * (fx0 - fx1 + fy0 - fy1);
      }
      o[idx] = v;
    }
  }
};

struct HeatApply {
  Grid g;
  int n_con, hydrostatic;
  double bdt, delt_max, cp_air, cv_air, rdg, k1k;
  double *pt, *hs;
  const double *delp, *delz;
  double *pkz;
  const double *cappa;  // A x npz (thermostruct%moist_kappa, dyn_core.F90:1338-1340) or null
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int k = bz + 1, n = g.nx * g.ny;
    if (k > n_con) return;
    double delt = fabs(bdt * delt_max);
    if (!hydrostatic) {
      if (k == 1) delt = 0.1 * delt;
      if (k == 2) delt = 0.5 * delt;
    }
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const size_t o = (size_t)bz * g.nA() + g.iA(i, j), c = (size_t)bz * g.nCC() + idx;
      if (hydrostatic) {
        if (k < 3) {
          pt[o] = pt[o] + hs[o] / (cp_air * delp[o] * pkz[c]);
        } else {
          const double dtmp = hs[o] / (cp_air * delp[o]);
          pt[o] = pt[o] + fsign(dmin(fabs(bdt) * delt_max, fabs(dtmp)), dtmp) / pkz[c];
          hs[o] = dtmp;
        }
      } else {
        const double ex = cappa ? cappa[o] / (1. - cappa[o]) : k1k;
        const double pz = dexp(ex * dlog(rdg * delp[o] / delz[c] * pt[o]));
        pkz[c] = pz;
        const double dtmp = hs[o] / (cv_air * delp[o]);
        pt[o] = pt[o] + fsign(dmin(delt, fabs(dtmp)), dtmp) / pz;
        hs[o] = dtmp;
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// hydrostatic pressure gradient: external-mode divergence coefficient and one_grad_p on corner values
struct Divg2Ext {  // dyn_core.F90:745-747, :791-797, :828-848
  Grid g;
  int npz;
  double d2_divg;
  const double *delp, *vt;
  double *divg2;  // A kind 2-D, corner indices
  CubedGeom cg;   // cubed sphere: the edge weights of a2b_ord2
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int w = g.nx + 1, ncol = w * (g.ny + 1);
    const size_t nA = g.nA();
    const bool cubed = g.grid_type < 3;
    const int npx = g.npx, npy = g.npy;
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % w, j = g.js + c / w;
      const int o = g.iA(i, j), o00 = g.iA(i - 1, j - 1), o10 = g.iA(i, j - 1), o01 = g.iA(i - 1, j);
      double wk = 0., d2 = 0.;
      for (int k = 0; k < npz; k++) {
        const double *dp = delp + (size_t)k * nA;
        double ptc;  // a2b_ord2 of delp at corner (i, j), a2b_edge.F90:329-450
        if (!cubed || (i > 1 && i < npx && j > 1 && j < npy)) {
          ptc = 0.25 * (dp[o00] + dp[o10] + dp[o01] + dp[o]);  // :377 / :427-433
        } else {
          constexpr double r3 = 1. / 3.;
          auto DP = [&](int ii, int jj) { return dp[g.iA(ii, jj)]; };
          if (i == 1 && j == 1)
            ptc = r3 * (DP(1, 1) + DP(1, 0) + DP(0, 1));  // :382-385
          else if (i == npx && j == 1)
            ptc = r3 * (DP(npx - 1, 1) + DP(npx - 1, 0) + DP(npx, 1));
          else if (i == npx && j == npy)
            ptc = r3 * (DP(npx - 1, npy - 1) + DP(npx, npy - 1) + DP(npx - 1, npy));
          else if (i == 1 && j == npy)
            ptc = r3 * (DP(1, npy - 1) + DP(0, npy - 1) + DP(1, npy));
          else if (i == 1 || i == npx) {  // :388-405
            const int ia = (i == 1) ? 0 : npx - 1;
            const double ew = (i == 1) ? cg.edge_w[j] : cg.edge_e[j];
            const double qa = 0.5 * (DP(ia, j - 1) + DP(ia + 1, j - 1)), qb = 0.5 * (DP(ia, j) + DP(ia + 1, j));
            ptc = ew * qa + (1. - ew) * qb;
          } else {  // :408-425
            const int ja = (j == 1) ? 0 : npy - 1;
            const double es = (j == 1) ? cg.edge_s[i] : cg.edge_n[i];
            const double qa = 0.5 * (DP(i - 1, ja) + DP(i - 1, ja + 1)), qb = 0.5 * (DP(i, ja) + DP(i, ja + 1));
            ptc = es * qa + (1. - es) * qb;
          }
        }
        if (k == 0) {
          wk = ptc;
          d2 = wk * vt[o];
        } else {
          wk = wk + ptc;
          d2 = d2 + ptc * vt[(size_t)k * nA + o];
        }
      }
      divg2[o] = d2_divg * d2 / wk;
    }
  }
};

// adv_pe (dyn_core.F90:1529-1632) on a cubed-sphere face.  (1) pem(k) = ptop + sum_{m<k} delp_before(m) on (is-1:ie+1, js-1:je+1),
// levels 1..npz+1 of an A slab; (2) one thread per cell and level: corner pressures by a2b_ord2 (a2b_edge.F90:329-425), the
// Green's-theorem gradient projected on the wind at the level's lower interface.
struct PemColumns {
  Grid g;
  int km;
  double ptop;
  const double *delp;
  double *pem;  // A x (km+1)
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int w = g.nx + 2, ncol = w * (g.ny + 2);
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is - 1 + c % w, j = g.js - 1 + c / w;
      const int o = g.iA(i, j);
      double p = ptop;
      pem[o] = p;
      for (int k = 1; k <= km; k++) {
        p = p + delp[(size_t)(k - 1) * nA + o];
        pem[(size_t)k * nA + o] = p;
      }
    }
  }
};

// (2a) the corner pressures pb(is:ie+1, js:je+1) of every level into an A slab (each corner is shared by four cells);
// (2b) the projection, reading them back
struct AdvPeCorners {
  Grid g;
  CubedGeom cg;
  const double *pem;
  double *pb;  // A x km: pb(:, :, k) = a2b_ord2(pem(:, k+1, :))
  static constexpr int CH = 1024;
  FV3_HD double corner(const double *pin, int i, int j) const {  // a2b_ord2, grid_type < 3, not a bounded domain
    const int npx = g.npx, npy = g.npy;
    auto Q = [&](int ii, int jj) { return pin[g.iA(ii, jj)]; };
    if (i > 1 && i < npx && j > 1 && j < npy) return 0.25 * (Q(i - 1, j - 1) + Q(i, j - 1) + Q(i - 1, j) + Q(i, j));
    constexpr double r3 = 1. / 3.;
    if (i == 1 && j == 1) return r3 * (Q(1, 1) + Q(1, 0) + Q(0, 1));
    if (i == npx && j == 1) return r3 * (Q(npx - 1, 1) + Q(npx - 1, 0) + Q(npx, 1));
    if (i == npx && j == npy) return r3 * (Q(npx - 1, npy - 1) + Q(npx, npy - 1) + Q(npx - 1, npy));
    if (i == 1 && j == npy) return r3 * (Q(1, npy - 1) + Q(0, npy - 1) + Q(1, npy));
    if (i == 1 || i == npx) {
      const int ia = (i == 1) ? 0 : npx - 1;
      const double ew = (i == 1) ? cg.edge_w[j] : cg.edge_e[j];
      const double qa = 0.5 * (Q(ia, j - 1) + Q(ia + 1, j - 1)), qb = 0.5 * (Q(ia, j) + Q(ia + 1, j));
      return ew * qa + (1. - ew) * qb;
    }
    const int ja = (j == 1) ? 0 : npy - 1;
    const double es = (j == 1) ? cg.edge_s[i] : cg.edge_n[i];
    const double qa = 0.5 * (Q(i - 1, ja) + Q(i - 1, ja + 1)), qb = 0.5 * (Q(i, ja) + Q(i, ja + 1));
    return es * qa + (1. - es) * qb;
  }
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int w = g.nx + 1, n = w * (g.ny + 1);
    const size_t nA = g.nA();
    const double *pin = pem + (size_t)(bz + 1) * nA;  // pem(:, k+1, :)
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % w, j = g.js + idx / w;
      pb[(size_t)bz * nA + g.iA(i, j)] = corner(pin, i, j);
    }
  }
};

struct AdvPe {
  Grid g;
  CubedGeom cg;
  int km;
  const double *ua, *va, *pb;
  double *om;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int k = bz + 1, n = g.nx * g.ny;
    const size_t nA = g.nA(), nFX = g.nFX(), nFY = g.nFY();
    const double *pc = pb + (size_t)(k - 1) * nA;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const size_t o = g.iA(i, j), o3 = (size_t)(k - 1) * nA + o;
      const double up = (k == km) ? ua[o3] : 0.5 * (ua[o3] + ua[o3 + nA]);
      const double vp = (k == km) ? va[o3] : 0.5 * (va[o3] + va[o3 + nA]);
      const double p00 = pc[o], p10 = pc[g.iA(i + 1, j)], p01 = pc[g.iA(i, j + 1)], p11 = pc[g.iA(i + 1, j + 1)];
      const double dxs = g.dx[g.iU(i, j)], dxn = g.dx[g.iU(i, j + 1)], dyw = g.dy[g.iV(i, j)], dye = g.dy[g.iV(i + 1, j)];
      double dot = 0.;
      for (int m = 0; m < 3; m++) {
        const double v3 = up * cg.ec1[(size_t)m * nA + o] + vp * cg.ec2[(size_t)m * nA + o];
        const double pdx_s = (p00 + p10) * dxs * cg.en1[(size_t)m * nFY + g.iFY(i, j)];
        const double pdx_n = (p01 + p11) * dxn * cg.en1[(size_t)m * nFY + g.iFY(i, j + 1)];
        const double pdy_w = (p00 + p01) * dyw * cg.en2[(size_t)m * nFX + g.iFX(i, j)];
        const double pdy_e = (p10 + p11) * dye * cg.en2[(size_t)m * nFX + g.iFX(i + 1, j)];
        const double grad = pdx_n - pdx_s - pdy_w + pdy_e;
        dot = (m == 0) ? v3 * grad : dot + v3 * grad;
      }
      om[o3] = om[o3] + 0.5 * g.rarea[o] * dot;
    }
  }
};

struct OneGradPHydro {  // dyn_core.F90:2002-2028 on precomputed corner values of pk, gz; with du / dv: grad1_p_update :2033-2116
  Grid g;
  double dt;
  const double *pk, *gz;   // corner slabs, npz+1 levels
  const double *divg2;     // null = no external-mode damping
  double *u, *v;
  double beta = 0.;
  double *du = nullptr, *dv = nullptr;
  const double *dpc = nullptr;   // the nonhydrostatic form (hydrostatic = .false., :1996-1997): a2b_ord4 of delp, npz corner slabs, as wk
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int k = bz;
    const size_t nA = g.nA();
    const double *pk0 = pk + (size_t)k * nA, *pk1 = pk0 + nA, *gz0 = gz + (size_t)k * nA, *gz1 = gz0 + nA;
    const double *dp = dpc ? dpc + (size_t)k * nA : nullptr;
    const int w = g.nx + 1, n = w * (g.ny + 1);
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % w, j = g.js + idx / w;
      const int o = g.iA(i, j), oe = g.iA(i + 1, j), on = g.iA(i, j + 1);
      const double wk0 = dp ? dp[o] : pk1[o] - pk0[o];
      if (i <= g.ie) {
        const double wke = dp ? dp[oe] : pk1[oe] - pk0[oe];
        const double wk2 = divg2 ? divg2[o] - divg2[oe] : 0.;
        double *p = u + (size_t)k * g.nU() + g.iU(i, j);
        if (du) {
          double *q = du + (size_t)k * g.nU() + g.iU(i, j);
          const double u0 = *p + beta * *q;                                         // :2098
          const double d1 = dt / (wk0 + wke) * ((gz1[o] - gz0[oe]) * (pk1[oe] - pk0[o]) + (gz0[o] - gz1[oe]) * (pk1[o] - pk0[oe]));
          *q = d1;
          const double ud = divg2 ? u0 + divg2[o] - divg2[oe] : u0;                 // :2102, left to right
          *p = (ud + (1. - beta) * d1) * g.rdx[g.iU(i, j)];
        } else
        *p = g.rdx[g.iU(i, j)] * (wk2 + *p + dt / (wk0 + wke) * ((gz1[o] - gz0[oe]) * (pk1[oe] - pk0[o]) +
                                                                (gz0[o] - gz1[oe]) * (pk1[o] - pk0[oe])));
      }
      if (j <= g.je) {
        const double wkn = dp ? dp[on] : pk1[on] - pk0[on];
        const double wk1 = divg2 ? divg2[o] - divg2[on] : 0.;
        double *p = v + (size_t)k * g.nV() + g.iV(i, j);
        if (dv) {
          double *q = dv + (size_t)k * g.nV() + g.iV(i, j);
          const double v0 = *p + beta * *q;                                         // :2107
          const double d1 = dt / (wk0 + wkn) * ((gz1[o] - gz0[on]) * (pk1[on] - pk0[o]) + (gz0[o] - gz1[on]) * (pk1[o] - pk0[on]));
          *q = d1;
          const double vd = divg2 ? v0 + divg2[o] - divg2[on] : v0;                 // :2111
          *p = (vd + (1. - beta) * d1) * g.rdy[g.iV(i, j)];
        } else
        *p = g.rdy[g.iV(i, j)] * (wk1 + *p + dt / (wk0 + wkn) * ((gz1[o] - gz0[on]) * (pk1[on] - pk0[o]) +
                                                                (gz0[o] - gz1[on]) * (pk1[o] - pk0[on])));
      }
    }
  }
};

struct CopyAtoCC {  // compute-domain copy of an A-kind field into a CC-kind one (pk = pkc, dyn_core.F90:1001-1010)
  Grid g;
  const double *src;
  double *dst;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int n = g.nx * g.ny;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT)
      dst[(size_t)bz * g.nCC() + idx] = src[(size_t)bz * g.nA() + g.iA(g.is + idx % g.nx, g.js + idx / g.nx)];
  }
};

struct PtToThetaV {  // fv_dynamics.F90:296-329, :379-399
  Grid g;
  int hydrostatic;  // 1: pkz given; 0: pkz computed, pt converted; -1: pkz computed only (:323-326; Rayleigh_Friction follows)
  double zvir, kappa, rdg;
  double *pt;
  const double *delp, *delz, *qv;
  double *pkz;
  // moist thermodynamics (fv3_set_moist): moist_kappa -> cappa, q_con from moist_cv and pkz with them (:305-317);
  // use_cond -> the conversion carries (1 - q_con) (:381-388).  mq = &q(isd,jsd,1,1) when moist_kappa
  RemapPar mp;
  const double *mq;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int n = g.nx * g.ny;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const size_t o = (size_t)bz * g.nA() + g.iA(i, j), c = (size_t)bz * g.nCC() + idx;
      const double dp1 = qv ? zvir * qv[o] : 0.;
      double pz, qc = 0.;
      if (hydrostatic > 0) {
        pz = pkz[c];
        if (mp.use_cond) qc = mp.q_con[o];
      } else if (mp.moist_kappa) {
        const double cvm = moist_cv(mp, mq + o, g.nA() * g.npz, qc);
        const double cap = mp.rdgas / (mp.rdgas + cvm / (1. + dp1));
        mp.q_con[o] = qc;
        mp.cappa[o] = cap;
        pz = dexp(cap * dlog(rdg * delp[o] * pt[o] * (1. + dp1) * (1. - qc) / delz[c]));
        pkz[c] = pz;
      } else {
        pz = dexp(kappa * dlog(rdg * delp[o] * pt[o] * (1. + dp1) / delz[c]));
        pkz[c] = pz;
        if (mp.use_cond) qc = mp.q_con[o];
      }
      if (hydrostatic >= 0) pt[o] = mp.use_cond ? pt[o] * (1. + dp1) * (1. - qc) / pz : pt[o] * (1. + dp1) / pz;
    }
  }
};

// cubed_to_latlon on a Cartesian domain (grid_type >= 4): c2l_ord2 (fv_grid_utils.F90:2551-2558) or c2l_ord4 (:2468-2475;
// the halo of u, v must be current).  With u2f != nullptr also the squared wind speed of Rayleigh_Friction
// (fv_dynamics.F90:1190-1205).
struct C2L {
  Grid g;
  int ord, hydrostatic;
  const double *u, *v, *w;
  double *ua, *va, *u2f;
  CubedGeom cg;   // grid_type < 4: the cubed_to_latlon matrix a11 .. a22
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    constexpr double a1 = 0.5625, a2 = -0.0625;
    const int n = g.nx * g.ny;
    const double *uk = u + (size_t)bz * g.nU(), *vk = v + (size_t)bz * g.nV();
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const size_t o = (size_t)bz * g.nA() + g.iA(i, j);
      double a, b;
      if (g.grid_type < 4) {
        // a face of the cubed sphere (fv_grid_utils.F90:2384-2466 ord 4, :2526-2546 ord 2): the "vorticity-conserving" two-point
        // form in the rows / columns next to the face edges (and everywhere with ord 2), 4th order inside; then the rotation to
        // (east, north) with a11 .. a22 (which carry the factor 1/2 of the two-point sums)
        constexpr double c1 = 1.125, c2 = -0.125;
        double ut, vt;
        const bool edge = ord == 2 || i == 1 || i == g.npx - 1 || j == 1 || j == g.npy - 1;
        if (edge) {
          const double dx0 = g.dx[g.iU(i, j)], dx1 = g.dx[g.iU(i, j + 1)], dy0 = g.dy[g.iV(i, j)], dy1 = g.dy[g.iV(i + 1, j)];
          ut = 2. * (uk[g.iU(i, j)] * dx0 + uk[g.iU(i, j + 1)] * dx1) / (dx0 + dx1);
          vt = 2. * (vk[g.iV(i, j)] * dy0 + vk[g.iV(i + 1, j)] * dy1) / (dy0 + dy1);
        } else {
          ut = c2 * (uk[g.iU(i, j - 1)] + uk[g.iU(i, j + 2)]) + c1 * (uk[g.iU(i, j)] + uk[g.iU(i, j + 1)]);
          vt = c2 * (vk[g.iV(i - 1, j)] + vk[g.iV(i + 2, j)]) + c1 * (vk[g.iV(i, j)] + vk[g.iV(i + 1, j)]);
        }
        const int oa = g.iA(i, j);
        a = cg.a11[oa] * ut + cg.a12[oa] * vt;
        b = cg.a21[oa] * ut + cg.a22[oa] * vt;
      } else if (ord == 2) {
        a = 0.5 * (uk[g.iU(i, j)] + uk[g.iU(i, j + 1)]);
        b = 0.5 * (vk[g.iV(i, j)] + vk[g.iV(i + 1, j)]);
      } else {
        a = a2 * (uk[g.iU(i, j - 1)] + uk[g.iU(i, j + 2)]) + a1 * (uk[g.iU(i, j)] + uk[g.iU(i, j + 1)]);
        b = a2 * (vk[g.iV(i - 1, j)] + vk[g.iV(i + 2, j)]) + a1 * (vk[g.iV(i, j)] + vk[g.iV(i + 1, j)]);
      }
      ua[o] = a;
      va[o] = b;
      if (u2f) u2f[o] = hydrostatic ? a * a + b * b : a * a + b * b + w[o] * w[o];
    }
  }
};

// Rayleigh_Friction after the halo update of u2f (fv_dynamics.F90:1211-1260): frictional heating and the implicit
// damping of u, v, w on the levels above rf_cutoff.  The reference overwrites its local u2f with rf*sqrt(u2f/u000) on
// is-1:ie+1; here that value is formed where it is used and u2f stays an input.
struct RayleighApply {
  Grid g;
  int conserve, hydrostatic;
  double cp, rg, ptop;
  const double *pm, *rf;  // device, kmax
  const double *u2f;
  double *pt, *delz, *u, *v, *w;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    constexpr double u000 = 4900.;
    const int wdt = g.nx + 1, n = wdt * (g.ny + 1);
    const double rfk = rf[bz], rcv = 1. / (cp - rg);
    const double *f = u2f + (size_t)bz * g.nA();
    auto damp = [&](int i, int j) { return rfk * sqrt(f[g.iA(i, j)] / u000); };
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % wdt, j = g.js + idx / wdt;
      const bool in_i = i <= g.ie, in_j = j <= g.je;
      const double d0 = damp(i, j);
      if (in_i && in_j) {
        const size_t o = (size_t)bz * g.nA() + g.iA(i, j);
        if (conserve) {
          const double x = f[g.iA(i, j)];
          const double d = 1. + d0;
          if (hydrostatic) {
            pt[o] = pt[o] + 0.5 * x / (cp - rg * ptop / pm[bz]) * (1. - 1. / (d * d));
          } else {
            const size_t c = (size_t)bz * g.nCC() + g.iCC(i, j);
            const double dz = delz[c] / pt[o];
            const double t = pt[o] + 0.5 * x * rcv * (1. - 1. / (d * d));
            pt[o] = t;
            delz[c] = dz * t;
          }
        }
        if (!hydrostatic) w[o] = w[o] / (1. + d0);
      }
      if (in_i) {
        const size_t ou = (size_t)bz * g.nU() + g.iU(i, j);
        u[ou] = u[ou] / (1. + 0.5 * (damp(i, j - 1) + d0));
      }
      if (in_j) {
        const size_t ov = (size_t)bz * g.nV() + g.iV(i, j);
        v[ov] = v[ov] / (1. + 0.5 * (damp(i - 1, j) + d0));
      }
    }
  }
};

// Rayleigh_Super (fv_dynamics.F90:1056-1121): u2f(:,:,k) = 1 / (1 + rf(k)) is a level constant, 0.5*(u2f + u2f) = u2f exactly
struct RayleighSuper {
  Grid g;
  int conserve, hydrostatic;
  double cp, rg, ptop;
  const double *pm, *rf;  // device, kmax
  const double *ua, *va;
  double *pt, *u, *v, *w;
  const double *u00, *v00;  // is_ideal_case: the t = 0 winds, or null
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int wdt = g.nx + 1, n = wdt * (g.ny + 1);
    const double rfk = rf[bz], rcv = 1. / (cp - rg);
    const double u2f = 1. / (1. + rfk);
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % wdt, j = g.js + idx / wdt;
      const bool in_i = i <= g.ie, in_j = j <= g.je;
      const size_t o = (size_t)bz * g.nA() + g.iA(i, j), ou = (size_t)bz * g.nU() + g.iU(i, j), ov = (size_t)bz * g.nV() + g.iV(i, j);
      if (u00) {  // :1064-1081
        if (in_i && in_j && !hydrostatic) w[o] = w[o] / (1. + rfk);
        if (in_i) u[ou] = (u[ou] + rfk * u00[ou]) / (1. + rfk);
        if (in_j) v[ov] = (v[ov] + rfk * v00[ov]) / (1. + rfk);
        continue;
      }
      if (in_i && in_j) {
        if (conserve) {  // :1084-1098
          const double a = ua[o], b = va[o];
          if (hydrostatic) {
            pt[o] = pt[o] + 0.5 * (a * a + b * b) * (1. - u2f * u2f) / (cp - rg * ptop / pm[bz]);
          } else {
            const double ww = w[o];
            pt[o] = pt[o] + 0.5 * (a * a + b * b + ww * ww) * (1. - u2f * u2f) * rcv;
          }
        }
        if (!hydrostatic) w[o] = u2f * w[o];
      }
      if (in_i) u[ou] = 0.5 * (u2f + u2f) * u[ou];
      if (in_j) v[ov] = 0.5 * (u2f + u2f) * v[ov];
    }
  }
};

struct OmgaUpdate {  // dyn_core.F90:409-421, :1182-1191
  Grid g;
  int km;
  double rdt, ptop;
  const double *pe, *delp0;
  double *omga;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const int o = g.iA(i, j);
      const size_t peb = (size_t)(j - (g.js - 1)) * (g.nx + 2) * (km + 1) + (i - (g.is - 1));
      double pem = ptop;
      for (int k = 1; k <= km; k++) {
        pem = pem + delp0[(size_t)(k - 1) * nA + o];
        omga[(size_t)(k - 1) * nA + o] = (pe[peb + (size_t)k * (g.nx + 2)] - pem) * rdt;
      }
    }
  }
};

// Ray_fast, dyn_core.F90:2485-2601 (RF_fast: the "inline" Rayleigh friction at the end of every acoustic substep, :1057-1060): on the
// levels k <= kmax (pfull < rf_cutoff) u, v (and w) are scaled by rf(k) = 1 / (1 + rff(k)); what the winds of a column lose,
// sum (1 - rf(k)) dp(k) u(k) / dm, goes back to its levels k <= k_rf (dm = sum of dp over them).  One thread per (i, j) of
// [is, ie+1] x [js, je+1] takes the u column, the v column and the w column that start there; rf, dp: device tables (the profile and dm
// are the host's, evaluated once in the reference's order: fv3_set_ray_fast).
struct RayFast {
  Grid g;
  int kmax, k_rf, hydrostatic;
  double dm;
  const double *rf, *dp;
  double *u, *v, *w;
  FV3_HD void column(double *f, size_t ls) const {
    double dmf = 0.;
    for (int k = 0; k < kmax; k++) {
      const double x = f[(size_t)k * ls];
      dmf = dmf + (1. - rf[k]) * dp[k] * x;
      f[(size_t)k * ls] = rf[k] * x;
    }
    dmf = dmf / dm;
    for (int k = 0; k < k_rf; k++) f[(size_t)k * ls] = f[(size_t)k * ls] + dmf;
  }
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int wd = g.nx + 1, ncol = wd * (g.ny + 1);
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % wd, j = g.js + c / wd;
      if (i <= g.ie) column(u + g.iU(i, j), g.nU());
      if (j <= g.je) column(v + g.iV(i, j), g.nV());
      if (!hydrostatic && i <= g.ie && j <= g.je) {
        const size_t nA = g.nA();
        double *wc = w + g.iA(i, j);
        for (int k = 0; k < kmax; k++) wc[(size_t)k * nA] = rf[k] * wc[(size_t)k * nA];
      }
    }
  }
};

// mix_dp, dyn_core.F90:2119-2200 (flagstruct%fill_dp, called behind d_sw at :820 with CG = .false.): a layer whose delp fell below
// 1 % of its reference thickness (or is NaN: `.not. delp >= dpmin`) takes the missing mass from the layer below -- the bottom layer from
// the one above -- and mixes pt (and w) with it.  Sequential in k within a column (the layer that gave mass is tested next): one thread
// per column of the compute domain marching k with delp of the next layer in a register; pt and w are touched only where the fix acts.
// ak, bk: the context's device tables (fv3_set_ak_bk); dpmin with the reference's expression, bit for bit.
// `.not. delp >= dpmin` (dyn_core.F90:2163 "catches NaN"): the library is compiled with -fno-honor-nans, under which !(a >= b) becomes
// a < b and a NaN would pass (the option folds the class test v_cmp_class_f64 away as well): NaN is read off the bit pattern, an
// integer fact the option does not touch
FV3_HD bool mix_dp_thin(double d, double dpmin) {
  return (fv3m_bits(d) & 0x7fffffffffffffffLL) > 0x7ff0000000000000LL || d < dpmin;
}

struct MixDp {
  Grid g;
  int hydrostatic;
  const double *ak, *bk;
  double *w, *delp, *pt;
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny, km = g.npz;
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const size_t o = (size_t)g.iA(i, j);
      double d0 = delp[o];
      for (int k = 0; k < km - 1; k++) {
        const double dpmin = 0.01 * (ak[k + 1] - ak[k] + (bk[k + 1] - bk[k]) * 1.E5);
        double d1 = delp[(size_t)(k + 1) * nA + o];
        if (mix_dp_thin(d0, dpmin)) {
          const double dp = dpmin - d0;
          const size_t o0 = (size_t)k * nA + o, o1 = o0 + nA;
          pt[o0] = (pt[o0] * d0 + pt[o1] * dp) / dpmin;
          if (!hydrostatic) w[o0] = (w[o0] * d0 + w[o1] * dp) / dpmin;
          delp[o0] = dpmin;
          d1 = d1 - dp;
          delp[o1] = d1;
        }
        d0 = d1;
      }
      {   // bottom (k = km): from above
        const double dpmin = 0.01 * (ak[km] - ak[km - 1] + (bk[km] - bk[km - 1]) * 1.E5);
        if (mix_dp_thin(d0, dpmin)) {
          const double dp = dpmin - d0;
          const size_t o0 = (size_t)(km - 1) * nA + o, om = o0 - nA;
          pt[o0] = (pt[o0] * d0 + pt[om] * dp) / dpmin;
          if (!hydrostatic) w[o0] = (w[o0] * d0 + w[om] * dp) / dpmin;
          delp[o0] = dpmin;
          delp[om] = delp[om] - dp;
        }
      }
    }
  }
};

// compute_aam, fv_dynamics.F90:1266-1314 (after the caller's cubed_to_latlon, :1287): one thread per column of the compute domain
struct AamColumns {
  Grid g;
  double radius, omega, agrav, ptop;
  const double *coslat, *ua, *delp;   // coslat: A (2-D) = cos(agrid(:,:,2))
  double *aam, *m_fac, *ps;           // CC, CC, A (2-D)
  FV3_HD void operator()(int bx, int, int, int tid, double *) const {
    const int ncol = g.nx * g.ny;
    const size_t nA = g.nA();
    FV3_COL_FOR(c, ncol) {
      const int i = g.is + c % g.nx, j = g.js + c / g.nx;
      const int o = g.iA(i, j);
      const double r1 = radius * coslat[o], r2 = r1 * r1;
      double a = 0., m = 0., p = ptop;
      for (int k = 0; k < g.npz; k++) {
        double dm = delp[(size_t)k * nA + o];
        p = p + dm;
        dm = dm * agrav;
        a = a + (r2 * omega + r1 * ua[(size_t)k * nA + o]) * dm;
        m = m + dm * r2;
      }
      aam[g.iCC(i, j)] = a;
      m_fac[g.iCC(i, j)] = m;
      ps[o] = p;
    }
  }
};

// consv_am, fv_dynamics.F90:784-798: u += u00 l2c_u, v += u00 l2c_v; one thread per (i, j) of [is, ie+1] x [js, je+1] and level
struct ConsvAmApply {
  static constexpr int CH = 256;
  Grid g;
  double u00;
  const double *l2c_u, *l2c_v;   // U (2-D), V (2-D)
  double *u, *v;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int wd = g.nx + 1, ncol = wd * (g.ny + 1);
    for (int c = bx * CH + tid; c < (bx + 1) * CH && c < ncol; c += kNT) {
      const int i = g.is + c % wd, j = g.js + c / wd;
      if (i <= g.ie) u[(size_t)bz * g.nU() + g.iU(i, j)] = u[(size_t)bz * g.nU() + g.iU(i, j)] + u00 * l2c_u[g.iU(i, j)];
      if (j <= g.je) v[(size_t)bz * g.nV() + g.iV(i, j)] = v[(size_t)bz * g.nV() + g.iV(i, j)] + u00 * l2c_v[g.iV(i, j)];
    }
  }
};

}  // namespace fv3
