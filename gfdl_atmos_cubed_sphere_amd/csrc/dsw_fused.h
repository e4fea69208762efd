// dsw_fused.h -- the d_sw transports of delp, w and pt in ONE marching kernel (sw_core.F90:908-1066, :1249-1283).
//
// The per-field kernels of dsw_march.h re-read the Courant numbers / area fluxes for every field and pass the
// mass fluxes through a scratch pair: 272 B of HBM traffic per cell-update for the three transports.  Here one
// wavefront carries the three fv_tp_2d marches side by side, so every input row is read once and the mass fluxes
// stay in registers: delp, w, pt, crx, xfx, cry, yfx, mfx, mfy in; delp, w, pt, mfx, mfy (+ zeroed heat_source,
// diss_est) out = 128 B per cell-update (the algorithmic figure of SURVEY section 8d).  The price is registers:
// the three marches need ~100 live doubles per lane, so the kernel runs at one or two wavefronts per SIMD and
// relies on the three independent instruction streams (and one-row-ahead loads) for latency hiding.
// Used when hord_dp == hord_vt == hord_tm, use_cond = .false.; otherwise the per-field kernels run.
#pragma once

#include "dsw_march.h"

// 1: the row steps without control flow (spmd.h "branch-free rows"); 0 builds the general forms only (an A / B switch for variants)
#ifndef FV3_BF
#define FV3_BF 1
#endif
// 1: the uniform-metric momentum kernel under the register budget of three wavefronts per SIMD (157 VGPRs, no spill): 0.392 -> 0.369 ms
#ifndef FV3_MOM_3W
#define FV3_MOM_3W 1
#endif

namespace fv3 {

// what the three marches share at one step
struct Tp2dShared {
  vd cx, xf, ar, rax;   // row r: crx, xfx, area, ra_x = area + xfx(i) - xfx(i+1)
  vd cy, yf;            // face r-2: cry, yfx
  vd arj, cxj, ray;     // row r-3: area, crx, ra_y = area + yfx(j) - yfx(j+1)
  vd rrax, rray;        // RN(1/ra_x), RN(1/ra_y): the three fields divide by the same ra_x, ra_y (vdiv_r)
};

// per-field register state of one fv_tp_2d march (same pipeline as Tp2dState::step)
template <int HORD>
struct Tp2dField {
  static constexpr int ORD_IN = (HORD == 10) ? 8 : HORD;
  static constexpr int ORD_OU = HORD;
  PpmY<ORD_IN> ya;
  PpmY<ORD_OU> yb;
  vd fx2_0, fx2_1, fx2_2, fx2_3;
  vd fy2y_prev, fyv_prev;
  FV3_D void init() {
    ya.init();
    yb.init();
    fx2_0 = fx2_1 = fx2_2 = fx2_3 = vd(0.);
    fy2y_prev = fyv_prev = vd(0.);
  }
  FV3_D void step(const vd &qn, const Tp2dShared &sh, bool have_face, bool have_row, vd &fxv, vd &fyv0, vd &fyv1) {
    fx2_3 = fx2_2; fx2_2 = fx2_1; fx2_1 = fx2_0;
    fx2_0 = ppm_faces_x<ORD_IN>(qn, sh.cx);
    const vd t = sh.xf * fx2_0;
    const vd qj = vdiv_r(qn * sh.ar + t - shl1(t), sh.rax, sh.rrax);
    ya.push(qn);
    yb.push(qj);
    if (!have_face) return;
    const vd fy2 = ya.face(sh.cy);
    const vd fyo = yb.face(sh.cy);
    const vd fy2y = sh.yf * fy2;
    const vd fyv = 0.5 * (fyo + fy2);
    fyv1 = fyv;
    if (have_row) {
      const vd qi = vdiv_r(ya.row_m3() * sh.arj + fy2y_prev - fy2y, sh.ray, sh.rray);
      const vd fxo = ppm_faces_x<ORD_OU>(qi, sh.cxj);
      fxv = 0.5 * (fxo + fx2_3);
      fyv0 = fyv_prev;
    }
    fy2y_prev = fy2y;
    fyv_prev = fyv;
  }
};

// COURANT: the kernel also does DswCourant's work (sw_core.F90:850-902, :923-936): Courant numbers / area fluxes are
// formed from uc, vc on the fly, used, and stored for the later consumers (crx, xfx, cry, yfx; cx, cy accumulated).
// GM: geometry mode (Grid::geom); 2 = orthogonal and uniform: the metric terms are wave-uniform scalars and the
// sin_sg factors (= 1) drop out -- x*1 is exact, so the results are the ones of the general kernel
// FLUXES: see DswArgs::dfx -- the fluxes are stored instead of the updated fields
template <int HORD, bool NH, bool COURANT, int GM = 0, bool FLUXES = false>
struct DswTransportFused {
  static constexpr bool UNI = (GM == 2);
  // 330 registers with the metric rows (one wavefront per SIMD), 246-250 with uniform metrics (two: -Rpass-analysis=kernel-resource-usage).
  // Grouping the row steps (rotation by renaming) was measured: 256 + 26 AGPRs and one wavefront, or 22-26 spills at two -- slower
  // either way
  Grid g;
  DswArgs a;
  MarchDims md;

  struct In {  // everything one step reads
    vd dp, w, pt;          // row r
    vd ar, cx, xf;         // row r (COURANT: cx = uc row, xf unused)
    vd cy, yf;             // face r-2 (COURANT: cy = vc row)
    vd xfj, ra;            // row r-3: xfx, rarea (mfx, mfy, cx, cy are accumulated in L2: vaccum)
    vd ucj;                // UNI + COURANT: uc of row r-3
    // COURANT only: metric rows and the accumulators
    vd rdxa, dyr, sg3, sg1;            // row r: rdxa, dy, sin_sg(.,3), sin_sg(.,1)
    vd rdya0, rdya1, dxr, sg4, sg2;       // face r-2: rdya(j-1), rdya(j), dx, sin_sg(j-1,4), sin_sg(j,2)
  };

  FV3_D void operator()(int gid) const {
    if constexpr (FLUXES || !FV3_BF)
      run_general(gid);
    else
      run_bf(gid);
  }

  // ---- the row step without control flow (spmd.h "branch-free rows") ------------------------------------------------------------------
  // Every row of the segment, the warm-up rows included, runs the whole step: what the general form skips on the rows whose face / output
  // row does not exist yet is computed on values that are never kept, and the row conditions (the Courant rows / faces and the output row
  // are the segment's own) go into the stores -- `on` of vstore_b_nt -- instead of into branches.  What keeps a (wave-uniform) branch in
  // the loop are the four flux capacitors: they are masked L2 atomics under `if (on)` (an atomic that adds +0.0 on the rows that are
  // not the segment's own -- spmd.h vaccum_z -- was measured slower than the branch: every row would pay four atomics).  Around them
  // the compiler counts the stores in flight, so the wait for the prefetched rows is vmcnt(<stores of the step>) instead of vmcnt(0).
  // The one store that belongs to a single row of the whole tile (mfy of the north face of row je) follows the loop.
  FV3_D void run_bf(int gid) const {
    int strip, seg, kk, tjw;
    md.decode_tj(gid, strip, seg, kk, tjw);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * tjw;
    const int jB = (jA + tjw - 1 < g.je) ? jA + tjw - 1 : g.je;
    const int rlast = jB + 3;
    const int lFx1 = (ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY(), oCC = (size_t)k * g.nCC();
    const double *delp = a.delp + oA, *pt = a.pt + oA, *w = NH ? a.w + oA : nullptr;
    double *crx = a.crx + oCX, *xfx = a.xfx + oCX, *cry = a.cry + oCY, *yfx = a.yfx + oCY;
    double *mfx = a.mfx + oFX, *mfy = a.mfy + oFY;
    const int seg_last = (jB == g.je);
    const int rowA = (seg == 0) ? g.jsd : jA, rowB = seg_last ? g.jed : jB;
    const int lY0 = (strip == 0) ? s.lA0 : s.lC0, lY1 = (ilo + s.lC1 == g.ie) ? s.lA1 : s.lC1;
    const double dt = a.dt;
    const int mw = a.mask_w;
    const int oC0 = mw ? (s.lC0 > mw + 1 - ilo ? s.lC0 : mw + 1 - ilo) : s.lC0;
    const int oC1 = mw ? (s.lC1 < g.npx - mw - 1 - ilo ? s.lC1 : g.npx - mw - 1 - ilo) : s.lC1;
    const int oF1 = mw ? (lFx1 < g.npx - mw - 1 - ilo ? lFx1 : g.npx - mw - 1 - ilo) : lFx1;
    const int oJ0 = mw ? mw + 1 : g.jsd, oJ1 = mw ? g.npy - mw - 1 : g.jed + 1;
    const vm mCX = make_mask(s.lC0, lFx1), mCY = make_mask(lY0, lY1), mO = make_mask(oC0, oC1), mOF = make_mask(oC0, oF1),
             mC = make_mask(s.lC0, s.lC1);
    // a sponge level (dyn_core.F90:703-724: nord_w = 0, damp_w = d2_divg): the del-2 damping of w and its heating (sw_core.F90:950-982,
    // :1268-1274; del6_vt_flux :1608 with nord = 0) in the row step -- uniform metrics only; the host sends the level here only then
    const bool wdamp = UNI && NH && a.lv.damp_w[k] > 1.E-5;
    const double damp4 = wdamp ? a.lv.damp_w[k] * g.da_min_c : 0., dd8 = a.kgb * fabs(a.dt);

    auto load_in = [&](int r) {
      In in;
      const long iA = (long)g.iA(ilo, r), iCX = (long)g.iCX(ilo, r);
      in.dp = vload(delp, iA, s.A);
      in.pt = vload(pt, iA, s.A);
      in.w = NH ? vload(w, iA, s.A) : vd(0.);
      in.ar = UNI ? vd(g.c_area) : vload(g.area, iA, s.A);
      const int jf = (r - 2 < jA) ? jA : r - 2, j = (r - 3 < jA) ? jA : r - 3;
      const long iCY = (long)g.iCY(ilo, jf);
      if (COURANT) {
        const long nAp = (long)g.nA();
        in.cx = vload(a.uc + (size_t)k * g.nV(), (long)g.iV(ilo, r), s.A);
        const long iAf = (long)g.iA(ilo, jf), iAm = (long)g.iA(ilo, jf - 1), iUf = (long)g.iU(ilo, jf);
        in.cy = vload(a.vc + (size_t)k * g.nU(), iUf, s.A);
        if constexpr (!UNI) {
          in.rdxa = vload(g.rdxa, iA, s.A);
          in.dyr = vload(g.dy, (long)g.iV(ilo, r), s.A);
          in.sg3 = vload(g.sin_sg + 2 * nAp, iA, s.A);
          in.sg1 = vload(g.sin_sg, iA, s.A);
          in.rdya0 = vload(g.rdya, iAm, s.A);
          in.rdya1 = vload(g.rdya, iAf, s.A);
          in.dxr = vload(g.dx, iUf, s.A);
          in.sg4 = vload(g.sin_sg + 3 * nAp, iAm, s.A);
          in.sg2 = vload(g.sin_sg + nAp, iAf, s.A);
        }
      } else {
        in.cx = vload(crx, iCX, s.F);
        in.xf = vload(xfx, iCX, s.F);
        in.cy = vload(cry, iCY, s.A);
        in.yf = vload(yfx, iCY, s.A);
        in.xfj = vload(xfx, (long)g.iCX(ilo, j), s.F);
      }
      if constexpr (UNI && COURANT) {
        in.ucj = vload(a.uc + (size_t)k * g.nV(), (long)g.iV(ilo, j), s.A);  // uc of row r-3: its crx, xfx are re-formed
      }
      in.ra = UNI ? vd(g.c_rarea) : vload(g.rarea, (long)g.iA(ilo, j), s.C);
      return in;
    };

    Tp2dField<HORD> fd, fw, fp;  // delp, w, pt
    fd.init();
    fp.init();
    if (NH) fw.init();
    vd ar_1(1.), ar_2(1.), ar_3(1.), cx_1(0.), cx_2(0.), cx_3(0.);  // area / crx of rows r-1 .. r-3
    vd xf_1(0.), xf_2(0.), xf_3(0.);                                 // COURANT: xfx of rows r-1 .. r-3
    vd yf_prev(0.), fym_prev(0.);
    In nxt = load_in(jA - 3);
    vdrain_loads();
    for (int r = jA - 3; r <= rlast; r++) {
      const In in = nxt;
      const int j = r - 3, jf = r - 2;
      const int jc = j < jA ? jA : j, jfc = jf < jA ? jA : jf;   // rows of the (dropped) stores / zero additions of the warm-up steps
      // (the flux capacitors cx, cy, mfx, mfy, sw_core.F90:923-940, are L2 atomics below -- global_atomic_add_f64 without return: no
      // register for the old value, no load; every element is added to by one lane of one wavefront, so the sum is the plain IEEE one)
      nxt = load_in(r < rlast ? r + 1 : rlast);
      Tp2dShared sh;
      vd xfj = in.xfj, cxj_uni(0.);
      if (COURANT) {
        // x faces of row r (sw_core.F90:865, :882-888, :923-927)
        const vd x = dt * in.cx;
        const vb xpos = x > 0.;
        if constexpr (UNI) {
          sh.cx = x * g.c_rdxa;
          sh.xf = g.c_dy * x;
        } else {
          sh.cx = vsel(xpos, x * shr1(in.rdxa), x * in.rdxa);
          sh.xf = vsel(xpos, in.dyr * x * shr1(in.sg3), in.dyr * x * in.sg1);
        }
        {
          const bool on = r >= rowA && r <= rowB;
          const long iCX = (long)g.iCX(ilo, r);
          vstore_b_nt(crx, iCX, sh.cx, mCX, on);
          vstore_b_nt(xfx, iCX, sh.xf, mCX, on);
          // the flux capacitors cx, cy, mfx, mfy (sw_core.F90:923-940) stay L2 atomics under a lane mask: measured on one set of arrays
          // (tools/pair_ab2.py) 0.765 ms against 0.792 as load - add - store and 1.12 as unmasked atomics adding +0.0 on the masked lanes
          if (on) vaccum(a.cx + oCX, iCX, sh.cx, s.lC0, lFx1);
        }
        // y faces of row r-2 (:894-900, :933-936)
        const vd y = dt * in.cy;
        const vb ypos = y > 0.;
        if constexpr (UNI) {
          sh.cy = y * g.c_rdya;
          sh.yf = g.c_dx * y;
        } else {
          sh.cy = vsel(ypos, y * in.rdya0, y * in.rdya1);
          sh.yf = vsel(ypos, in.dxr * y * in.sg4, in.dxr * y * in.sg2);
        }
        {
          const bool on = jf >= jA && (jf <= jB || (seg_last && jf == g.je + 1));
          const long iCY = (long)g.iCY(ilo, jfc);
          vstore_b_nt(cry, iCY, sh.cy, mCY, on);
          vstore_b_nt(yfx, iCY, sh.yf, mCY, on);
          if (on) vaccum(a.cy + oCY, iCY, sh.cy, lY0, lY1);
        }
        if constexpr (UNI) {  // crx, xfx of row r-3 from its uc again (same expressions as at step r-3): no 3-row windows
          const vd xj = dt * in.ucj;
          xfj = g.c_dy * xj;
          cxj_uni = xj * g.c_rdxa;
        } else {
          xfj = xf_3;
          xf_3 = xf_2; xf_2 = xf_1; xf_1 = sh.xf;
        }
      } else {
        sh.cx = in.cx; sh.xf = in.xf;
        sh.cy = in.cy; sh.yf = in.yf;
      }
      sh.ar = in.ar;
      sh.rax = in.ar + sh.xf - shl1(sh.xf);
      sh.arj = UNI ? in.ar : ar_3;
      sh.cxj = (UNI && COURANT) ? cxj_uni : cx_3;
      sh.ray = sh.arj + yf_prev - sh.yf;
      sh.rrax = vrecip(sh.rax);
      sh.rray = vrecip(sh.ray);
      if constexpr (!UNI) { ar_3 = ar_2; ar_2 = ar_1; ar_1 = in.ar; }
      if constexpr (!(UNI && COURANT)) { cx_3 = cx_2; cx_2 = cx_1; cx_1 = sh.cx; }
      // (the warm-up steps skip the sweeps whose face / row does not exist yet: branches without a memory operation inside)
      const bool have_face = jf >= jA, have_row = j >= jA;
      vd fxd(0.), fyd0(0.), fyd1(0.), fxw(0.), fyw0(0.), fyw1(0.), fxp(0.), fyp0(0.), fyp1(0.);
      fd.step(in.dp, sh, have_face, have_row, fxd, fyd0, fyd1);
      if (NH) fw.step(in.w, sh, have_face, have_row, fxw, fyw0, fyw1);
      fp.step(in.pt, sh, have_face, have_row, fxp, fyp0, fyp1);
      // mass flux through y-face r-2 (tp_core.F90:222-226); carried to the next row as its south face
      const vd fym = fyd1 * sh.yf;
      {
        const bool on = j >= jA && j >= oJ0 && j <= oJ1;
        const vd fxm = fxd * xfj;  // tp_core.F90:217-221
        const vd fym0 = fym_prev, fym1 = fym;
        const long iFX = (long)g.iFX(ilo, jc), iFY0 = (long)g.iFY(ilo, jc), iA = (long)g.iA(ilo, jc), iCC = (long)g.iCC(ilo, jc);
        if (on) vaccum(mfx, iFX, fxm, oC0, oF1);
        if (on) vaccum(mfy, iFY0, fym0, oC0, oC1);
        const vd dp = fd.ya.row_m3();
        const vd dpn = dp + (fxm - shl1(fxm) + fym0 - fym1) * in.ra;
        vstore_b_nt(a.delp_out + oA, iA, dpn, mO, on);
        const vd rdpn = vrecip(dpn);
        {  // pt (sw_core.F90:1053-1066)
          const vd gx = fxp * fxm, gy0 = fyp0 * fym0, gy1 = fyp1 * fym1;
          vstore_b_nt(a.pt_out + oA, iA, vdiv_r(fp.ya.row_m3() * dp + (gx - shl1(gx) + gy0 - gy1) * in.ra, dpn, rdpn), mO, on);
        }
        vd hs(0.), de(0.);   // :943-948
        if (NH) {  // w (:985-989, :1262-1274)
          const vd gx = fxw * fxm, gy0 = fyw0 * fym0, gy1 = fyw1 * fym1;
          vd wn = vdiv_r(fw.ya.row_m3() * dp + (gx - shl1(gx) + gy0 - gy1) * in.ra, dpn, rdpn);
          if constexpr (UNI) {
            if (wdamp) {  // rows j-1, j, j+1 of w are rows r-4, r-3, r-2 of the window (no memory operation in this branch)
              const vd w0 = fw.ya.q1;
              const vd d2m = damp4 * fw.ya.q0, d20 = damp4 * w0, d2p = damp4 * fw.ya.q2;
              const vd fx2 = g.c_del6_v * (shr1(d20) - d20);
              const vd dw = (fx2 - shl1(fx2) + g.c_del6_u * (d2m - d20) - g.c_del6_u * (d20 - d2p)) * in.ra;
              const vd tmp = dw * (w0 + 0.5 * dw);   // 0.5 * [(w + dw)**2 - w**2]
              hs = g.prevent_diss_cooling ? dd8 - vmin(vd(0.), tmp) : dd8 - tmp;
              if (g.do_diss_est) de = dd8 - tmp;
              wn = wn + dw;
            }
          }
          vstore_b_nt(a.w_out + oA, iA, wn, mO, on);
        }
        // (NULL: the caller does not read them -- fv3_d_sw; the stores are dropped, their row pointer is any valid one)
        vstore_b_nt((a.heat_s ? a.heat_s : a.delp_out) + oCC, iCC, hs, mC, on && a.heat_s != nullptr && !a.skip_heat);
        vstore_b_nt((a.diss_e ? a.diss_e : a.delp_out) + oCC, iCC, de, mC, on && a.diss_e != nullptr && !a.skip_heat);
      }
      fym_prev = fym;
      yf_prev = sh.yf;
    }
    // the north face of the tile's north row: the last step (face je + 1) left its mass flux in fym_prev
    if (seg_last && g.je >= oJ0 && g.je <= oJ1) vaccum(mfy, (long)g.iFY(ilo, g.je + 1), fym_prev, oC0, oC1);
  }

  FV3_D void run_general(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int rlast = jB + 3;
    const int lFx1 = (ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY(), oCC = (size_t)k * g.nCC();
    const double *delp = a.delp + oA, *pt = a.pt + oA, *w = NH ? a.w + oA : nullptr;
    double *crx = a.crx + oCX, *xfx = a.xfx + oCX, *cry = a.cry + oCY, *yfx = a.yfx + oCY;
    double *mfx = a.mfx + oFX, *mfy = a.mfy + oFY;
    // COURANT: rows / columns of the Courant arrays this wavefront stores (the segment's own rows plus the halo rows
    // at the south / north end of the tile; the strip's own faces plus the halo columns at the west / east end)
    const int seg_last = (jB == g.je);
    const int rowA = (seg == 0) ? g.jsd : jA, rowB = seg_last ? g.jed : jB;
    const int lY0 = (strip == 0) ? s.lA0 : s.lC0, lY1 = (ilo + s.lC1 == g.ie) ? s.lA1 : s.lC1;
    const double dt = a.dt;
    // cubed-sphere hybrid: lanes / rows of the outputs this kernel owns (DswArgs::mask_w)
    const int mw = a.mask_w;
    const int oC0 = mw ? (s.lC0 > mw + 1 - ilo ? s.lC0 : mw + 1 - ilo) : s.lC0;
    const int oC1 = mw ? (s.lC1 < g.npx - mw - 1 - ilo ? s.lC1 : g.npx - mw - 1 - ilo) : s.lC1;
    const int oF1 = mw ? (lFx1 < g.npx - mw - 1 - ilo ? lFx1 : g.npx - mw - 1 - ilo) : lFx1;
    const int oJ0 = mw ? mw + 1 : g.jsd, oJ1 = mw ? g.npy - mw - 1 : g.jed + 1;

    auto load_in = [&](int r) {
      In in;
      const long iA = (long)g.iA(ilo, r), iCX = (long)g.iCX(ilo, r);
      in.dp = vload(delp, iA, s.A);
      in.pt = vload(pt, iA, s.A);
      in.w = NH ? vload(w, iA, s.A) : vd(0.);
      in.ar = UNI ? vd(g.c_area) : vload(g.area, iA, s.A);
      const int jf = (r - 2 < jA) ? jA : r - 2, j = (r - 3 < jA) ? jA : r - 3;
      const long iCY = (long)g.iCY(ilo, jf);
      if (COURANT) {
        const long nAp = (long)g.nA();
        in.cx = vload(a.uc + (size_t)k * g.nV(), (long)g.iV(ilo, r), s.A);
        const long iAf = (long)g.iA(ilo, jf), iAm = (long)g.iA(ilo, jf - 1), iUf = (long)g.iU(ilo, jf);
        in.cy = vload(a.vc + (size_t)k * g.nU(), iUf, s.A);
        if constexpr (!UNI) {
          in.rdxa = vload(g.rdxa, iA, s.A);
          in.dyr = vload(g.dy, (long)g.iV(ilo, r), s.A);
          in.sg3 = vload(g.sin_sg + 2 * nAp, iA, s.A);
          in.sg1 = vload(g.sin_sg, iA, s.A);
          in.rdya0 = vload(g.rdya, iAm, s.A);
          in.rdya1 = vload(g.rdya, iAf, s.A);
          in.dxr = vload(g.dx, iUf, s.A);
          in.sg4 = vload(g.sin_sg + 3 * nAp, iAm, s.A);
          in.sg2 = vload(g.sin_sg + nAp, iAf, s.A);
        }
      } else {
        in.cx = vload(crx, iCX, s.F);
        in.xf = vload(xfx, iCX, s.F);
        in.cy = vload(cry, iCY, s.A);
        in.yf = vload(yfx, iCY, s.A);
        in.xfj = vload(xfx, (long)g.iCX(ilo, j), s.F);
      }
      if constexpr (UNI && COURANT) {
        in.ucj = vload(a.uc + (size_t)k * g.nV(), (long)g.iV(ilo, j), s.A);  // uc of row r-3: its crx, xfx are re-formed
      }
      in.ra = UNI ? vd(g.c_rarea) : vload(g.rarea, (long)g.iA(ilo, j), s.C);
      return in;
    };

    Tp2dField<HORD> fd, fw, fp;  // delp, w, pt
    fd.init();
    fp.init();
    if (NH) fw.init();
    vd ar_1(1.), ar_2(1.), ar_3(1.), cx_1(0.), cx_2(0.), cx_3(0.);  // area / crx of rows r-1 .. r-3
    vd xf_1(0.), xf_2(0.), xf_3(0.);                                 // COURANT: xfx of rows r-1 .. r-3
    vd yf_prev(0.), fym_prev(0.);
    In nxt = load_in(jA - 3);
    for (int r = jA - 3; r <= rlast; r++) {
      const In in = nxt;
      nxt = load_in(r < rlast ? r + 1 : rlast);  // (loading two to four rows ahead was measured: no gain)
      const int j = r - 3;
      const bool have_face = r - 2 >= jA, have_row = j >= jA;
      Tp2dShared sh;
      vd xfj = in.xfj, cxj_uni(0.);
      if (COURANT) {
        // x faces of row r (sw_core.F90:865, :882-888, :923-927)
        const vd x = dt * in.cx;
        const vb xpos = x > 0.;
        if constexpr (UNI) {
          sh.cx = x * g.c_rdxa;
          sh.xf = g.c_dy * x;
        } else {
          sh.cx = vsel(xpos, x * shr1(in.rdxa), x * in.rdxa);
          sh.xf = vsel(xpos, in.dyr * x * shr1(in.sg3), in.dyr * x * in.sg1);
        }
        if (r >= rowA && r <= rowB) {
          const long iCX = (long)g.iCX(ilo, r);
          vstore_nt(crx, iCX, sh.cx, s.lC0, lFx1);
          vstore_nt(xfx, iCX, sh.xf, s.lC0, lFx1);
          vaccum(a.cx + oCX, iCX, sh.cx, s.lC0, lFx1);
        }
        // y faces of row r-2 (:894-900, :933-936)
        const vd y = dt * in.cy;
        const vb ypos = y > 0.;
        if constexpr (UNI) {
          sh.cy = y * g.c_rdya;
          sh.yf = g.c_dx * y;
        } else {
          sh.cy = vsel(ypos, y * in.rdya0, y * in.rdya1);
          sh.yf = vsel(ypos, in.dxr * y * in.sg4, in.dxr * y * in.sg2);
        }
        const int jf = r - 2;
        if (jf >= jA && (jf <= jB || (seg_last && jf == g.je + 1))) {
          const long iCY = (long)g.iCY(ilo, jf);
          vstore_nt(cry, iCY, sh.cy, lY0, lY1);
          vstore_nt(yfx, iCY, sh.yf, lY0, lY1);
          vaccum(a.cy + oCY, iCY, sh.cy, lY0, lY1);
        }
        if constexpr (UNI) {  // crx, xfx of row r-3 from its uc again (same expressions as at step r-3): no 3-row windows
          const vd xj = dt * in.ucj;
          xfj = g.c_dy * xj;
          cxj_uni = xj * g.c_rdxa;
        } else {
          xfj = xf_3;
          xf_3 = xf_2; xf_2 = xf_1; xf_1 = sh.xf;
        }
      } else {
        sh.cx = in.cx; sh.xf = in.xf;
        sh.cy = in.cy; sh.yf = in.yf;
      }
      sh.ar = in.ar;
      sh.rax = in.ar + sh.xf - shl1(sh.xf);
      sh.arj = UNI ? in.ar : ar_3;
      sh.cxj = (UNI && COURANT) ? cxj_uni : cx_3;
      sh.ray = sh.arj + yf_prev - sh.yf;
      sh.rrax = vrecip(sh.rax);
      sh.rray = vrecip(sh.ray);
      if constexpr (!UNI) { ar_3 = ar_2; ar_2 = ar_1; ar_1 = in.ar; }
      if constexpr (!(UNI && COURANT)) { cx_3 = cx_2; cx_2 = cx_1; cx_1 = sh.cx; }
      vd fxd, fyd0, fyd1, fxw, fyw0, fyw1, fxp, fyp0, fyp1;
      fd.step(in.dp, sh, have_face, have_row, fxd, fyd0, fyd1);
      if (NH) fw.step(in.w, sh, have_face, have_row, fxw, fyw0, fyw1);
      fp.step(in.pt, sh, have_face, have_row, fxp, fyp0, fyp1);
      if (have_face) {
        // mass flux through y-face r-2 (tp_core.F90:222-226); carried to the next row as its south face
        vd fym = fyd1 * sh.yf;
        const bool dk = FLUXES && a.dfx && a.dcoef[k] > 1.E-4;
        if (FLUXES && dk) fym = fym + vload(a.dfy + (size_t)k * g.nU(), (long)g.iU(ilo, r - 2), s.A);   // tp_core.F90:228-232
        if constexpr (FLUXES) {
          if (have_row && j >= oJ0 && j <= oJ1) {
            vd fxm = fxd * xfj;
            if (dk) fxm = fxm + vload(a.dfx + (size_t)k * g.nV(), (long)g.iV(ilo, j), s.A);
            const vd fym0 = fym_prev, fym1 = fym;
            const long iFX = (long)g.iFX(ilo, j), iFY0 = (long)g.iFY(ilo, j), iFY1 = (long)g.iFY(ilo, j + 1);
            const bool top = j == g.je;
            vstore(a.ofx + oFX, iFX, fxm, oC0, oF1);
            vstore(a.ofy + oFY, iFY0, fym0, oC0, oC1);
            if (top) vstore(a.ofy + oFY, iFY1, fym1, oC0, oC1);
            vstore(a.ogx + oFX, iFX, fxp * fxm, oC0, oF1);
            vstore(a.ogy + oFY, iFY0, fyp0 * fym0, oC0, oC1);
            if (top) vstore(a.ogy + oFY, iFY1, fyp1 * fym1, oC0, oC1);
            if (NH) {
              vstore(a.ogxw + oFX, iFX, fxw * fxm, oC0, oF1);
              vstore(a.ogyw + oFY, iFY0, fyw0 * fym0, oC0, oC1);
              if (top) vstore(a.ogyw + oFY, iFY1, fyw1 * fym1, oC0, oC1);
            }
          }
        } else
        if (have_row && j >= oJ0 && j <= oJ1) {
          const vd fxm = fxd * xfj;  // tp_core.F90:217-221
          const vd fym0 = fym_prev, fym1 = fym;
          const long iFX = (long)g.iFX(ilo, j), iFY0 = (long)g.iFY(ilo, j), iA = (long)g.iA(ilo, j);
          vaccum(mfx, iFX, fxm, oC0, oF1);            // sw_core.F90:928-940
          vaccum(mfy, iFY0, fym0, oC0, oC1);
          if (j == g.je) {
            const long iFY1 = (long)g.iFY(ilo, j + 1);
            vstore_nt(mfy, iFY1, vload(mfy, iFY1, s.C) + fym1, oC0, oC1);
          }
          const vd dp = fd.ya.row_m3();
          const vd dpn = dp + (fxm - shl1(fxm) + fym0 - fym1) * in.ra;
          vstore_nt(a.delp_out + oA, iA, dpn, oC0, oC1);
          const vd rdpn = vrecip(dpn);
          {  // pt (sw_core.F90:1053-1066)
            const vd gx = fxp * fxm, gy0 = fyp0 * fym0, gy1 = fyp1 * fym1;
            vstore_nt(a.pt_out + oA, iA, vdiv_r(fp.ya.row_m3() * dp + (gx - shl1(gx) + gy0 - gy1) * in.ra, dpn, rdpn), oC0, oC1);
          }
          if (NH) {  // w (:985-989, :1262-1274)
            const vd gx = fxw * fxm, gy0 = fyw0 * fym0, gy1 = fyw1 * fym1;
            vstore_nt(a.w_out + oA, iA, vdiv_r(fw.ya.row_m3() * dp + (gx - shl1(gx) + gy0 - gy1) * in.ra, dpn, rdpn), oC0, oC1);
          }
          const long iCC = (long)g.iCC(ilo, j);  // :943-948
          vstore_nt(a.heat_s + oCC, iCC, vd(0.), s.lC0, s.lC1);
          vstore_nt(a.diss_e + oCC, iCC, vd(0.), s.lC0, s.lC1);
        }
        fym_prev = fym;
      }
      if (have_face) yf_prev = sh.yf;
    }
  }
};

}  // namespace fv3

// =====================================================================================================
// DswMomentumFused: DswKeMarch + DswVortMarch in one marching kernel (same level conditions).  The KE / damping
// term of corner rows j and j+1 is formed in registers right when the wind update of row j needs it, so the ke
// scratch round trip and the second pass over u, v, crx, xfx, cry, yfx disappear:
// u, v, uc, vc, divg_d, crx, xfx, cry, yfx in; u, v, delpc out = 96 B per cell-update (was 136).
namespace fv3 {

// CS: the interior of a cubed-sphere face (hybrid of fv3_api.hip dsw_cubed: store mask, non-orthogonal B-grid winds); a
// compile-time switch so that the doubly periodic instantiations keep their register budget
template <int SWC, int HORD, int GM = 0, bool CS = false>
struct DswMomentumFused {
  static constexpr bool UNI = (GM == 2);  // orthogonal + uniform metrics: scalars instead of metric rows
  // two wavefronts per SIMD: 254 VGPRs with the general metric rows (the row-j values of the wind update are re-read at
  // the top of the step instead of being carried for three steps), 188 with uniform metrics; measured 0.735 -> 0.57 -> 0.49 ms
  static constexpr int kTwoWavesPerSimd = 1;
  // uniform metrics, branch-free form: 157 VGPRs -- three wavefronts per SIMD
  static constexpr int kThreeWavesPerSimd = (UNI && !CS && FV3_BF) ? FV3_MOM_3W : 0;
  Grid g;
  DswArgs a;
  MarchDims md;

  struct In {
    vd u1, v0, v1, dx1, dy0, dy1, ra, f0;          // vorticity of row r: u(r+1), v(r), v(r)@i+1, dx(r+1), dy(r), dy@i+1, rarea, f0
    vd ar, cx, xf, cy, yf;                         // fv_tp_2d: area, crx, xfx of row r; cry, yfx of face r-2
    vd vc, uc, ukc, rdy, rdx, dv, dgu, dgv, rac;   // corner row jc = r-2 (dv = divg_d(jc+1))
  };

  FV3_D void operator()(int gid) const {
    if constexpr (FV3_BF)
      run_bf(gid);
    else
      run_general(gid);
  }

  // ---- the row step without control flow (see DswTransportFused::run_bf) -------------------------------------------------------------------
  FV3_D void run_bf(int gid) const {
    int strip, seg, kk, tjw;
    md.decode_tj(gid, strip, seg, kk, tjw);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * tjw;
    const int jB = (jA + tjw - 1 < g.je) ? jA + tjw - 1 : g.je;
    const int rlast = jB + 3;
    const int lFx1 = (ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const double *u = a.u + (size_t)k * g.nU(), *v = a.v + (size_t)k * g.nV();
    const double *uc = a.uc + (size_t)k * g.nV(), *vc = a.vc + (size_t)k * g.nU();
    const double *dv = a.divg_d + (size_t)k * g.nB();
    const double *crx = a.crx + (size_t)k * g.nCX(), *xfx = a.xfx + (size_t)k * g.nCX();
    const double *cry = a.cry + (size_t)k * g.nCY(), *yfx = a.yfx + (size_t)k * g.nCY();
    double *uo = a.u_out + (size_t)k * g.nU(), *vo = a.v_out + (size_t)k * g.nV();
    double *dpc = a.delpc ? a.delpc + (size_t)k * g.nA() : nullptr;
    const double dt5 = 0.5 * a.dt, dt = a.dt;
    const int mw = CS ? a.mask_w : 0;
    const int oC0 = (CS && mw) ? (s.lC0 > mw + 1 - ilo ? s.lC0 : mw + 1 - ilo) : s.lC0;
    const int oC1 = (CS && mw) ? (s.lC1 < g.npx - mw - 1 - ilo ? s.lC1 : g.npx - mw - 1 - ilo) : s.lC1;
    const int oF1 = (CS && mw) ? (lFx1 < g.npx - mw - 1 - ilo ? lFx1 : g.npx - mw - 1 - ilo) : lFx1;
    const int oJ0 = (CS && mw) ? mw + 1 : g.jsd, oJ1 = (CS && mw) ? g.npy - mw - 1 : g.jed + 1;
    const double d2_bg = a.lv.d2_divg[k];
    const double damp2 = g.da_min_c * dmax(d2_bg, dmin(0.20, a.dddmp * 0.));            // :1454 with vort = 0
    const double dd8 = g.stretched_grid ? g.da_min * ipow(a.d4_bg, 2) : ipow(g.da_min_c * a.d4_bg, 2);  // :1446-1450
    const vm mO = make_mask(oC0, oC1), mOF = make_mask(oC0, oF1);
    // a sponge level (dyn_core.F90:703-724: nord_k = 0): the del-2 divergence damping, with the divergence formed from the D-grid winds
    // (sw_core.F90:1290-1371) -- on an orthogonal, uniform grid ptc = u * dyc and vort = v * dxc exactly (cosa = 0, sina = 1)
    const bool nord0 = UNI && a.lv.nord_k[k] == 0;

    auto load_in = [&](int r) {
      In in;
      const long oU1 = (long)g.iU(ilo, r + 1), oV = (long)g.iV(ilo, r), oA = (long)g.iA(ilo, r);
      in.u1 = vload(u, oU1, s.A);
      in.v0 = vload(v, oV, s.A);
      in.v1 = vload(v, oV + 1, s.A);  // V kind has the extra column ied+1
      in.f0 = vload(g.f0, oA, s.A);
      if constexpr (UNI) {
        in.dx1 = vd(g.c_dx);  in.dy0 = in.dy1 = vd(g.c_dy);
        in.ra = vd(g.c_rarea);  in.ar = vd(g.c_area);
      } else {
        in.dx1 = vload(g.dx, oU1, s.A);
        in.dy0 = vload(g.dy, oV, s.A);  in.dy1 = vload(g.dy, oV + 1, s.A);
        in.ra = vload(g.rarea, oA, s.A);
        in.ar = vload(g.area, oA, s.A);
      }
      if constexpr (UNI) {
        // uniform metrics: the Courant numbers / area fluxes are re-formed from uc, vc with the expressions of the transport kernel
        // (crx = dt uc rdxa, xfx = dy dt uc, ...: bit for bit what it stored) -- uc of row r is the one new row, vc of face r-2 is the
        // corner row's, uc of row r-3 the carried one: five row loads less per step
        in.cx = vload(uc, (long)g.iV(ilo, r), s.A);
      } else {
        const long oCX = (long)g.iCX(ilo, r);
        in.cx = vload(crx, oCX, s.F);
        in.xf = vload(xfx, oCX, s.F);
        const int jf = (r - 2 < jA) ? jA : r - 2;
        const long oCY = (long)g.iCY(ilo, jf);
        in.cy = vload(cry, oCY, s.A);
        in.yf = vload(yfx, oCY, s.A);
      }
      const int jc = (r - 2 < jA - 1) ? jA - 1 : r - 2;
      const long oUc = (long)g.iU(ilo, jc), oVc = (long)g.iV(ilo, jc), oBc = (long)g.iB(ilo, jc);
      in.vc = vload(vc, oUc, s.A);
      in.uc = vload(uc, oVc, s.A);
      in.ukc = vload(u, oUc, s.A);
      in.dv = vload(dv, (long)g.iB(ilo, jc + 1), s.A);
      if constexpr (UNI) {
        in.rdy = vd(g.c_rdy);  in.rdx = vd(g.c_rdx);
        in.dgu = vd(g.c_divg_u);  in.dgv = vd(g.c_divg_v);  in.rac = vd(g.c_rarea_c);
      } else {
        in.rdy = vload(g.rdy, oVc, s.A);
        in.rdx = vload(g.rdx, oUc, s.A);
        in.dgu = vload(g.divg_u, oUc, s.A);
        in.dgv = vload(g.divg_v, oVc, s.A);
        in.rac = vload(g.rarea_c, oBc, s.A);
      }
      return in;
    };

    Tp2dState<HORD, UNI> st;
    st.init();
    PpmYsw<SWC> yv;
    yv.init();
    vd vtdx_n(0.);
    vd uc_p(0.), rdy_p(0.), dgv_p(0.), d_m(0.), d_0(0.);
    vd ke_p(0.), yf_p(0.), fyv1_last(0.);
    {
      const long oU = (long)g.iU(ilo, jA - 3);
      vtdx_n = vload(u, oU, s.A) * (UNI ? vd(g.c_dx) : vload(g.dx, oU, s.A));
      d_0 = vload(dv, (long)g.iB(ilo, jA - 1), s.A);
    }
    In nxt = load_in(jA - 3);
    vdrain_loads();
    for (int r = jA - 3; r <= rlast; r++) {
      const In in = nxt;
      const int j = r - 3, jc = r - 2;
      const int jw = j >= jA ? j : jA;
      const long oUj = (long)g.iU(ilo, jw), oVj = (long)g.iV(ilo, jw);
      const vd u_j = vload(u, oUj, s.A), dx_j = UNI ? vd(g.c_dx) : vload(g.dx, oUj, s.A);
      const vd v_j = vload(v, oVj, s.A), dy_j = UNI ? vd(g.c_dy) : vload(g.dy, oVj, s.A);
      vd xf_j;
      if constexpr (UNI) xf_j = g.c_dy * (dt * uc_p);   // uc_p = uc of row r-3 once the pipeline is full
      else xf_j = vload(xfx, (long)g.iCX(ilo, jw), s.F);
      nxt = load_in(r < rlast ? r + 1 : rlast);
      // ---- absolute vorticity of row r (sw_core.F90:1231-1247, :1476-1495) -> fv_tp_2d march ---------------------------
      const vd vt0 = vtdx_n, vt1 = in.u1 * in.dx1, ut0 = in.v0 * in.dy0, ut1 = in.v1 * in.dy1;
      MarchIn mi;
      mi.qn = in.ra * (vt0 - vt1 - ut0 + ut1) + in.f0;
      mi.ar = in.ar;
      vd yf_r = in.yf;
      if constexpr (UNI) {
        const vd x = dt * in.cx, y = dt * in.vc;   // sw_core.F90:865-900 as in DswTransportFused
        mi.cx = x * g.c_rdxa; mi.xf = g.c_dy * x;
        mi.cy = y * g.c_rdya; mi.yf = g.c_dx * y;
        yf_r = mi.yf;
      } else {
        mi.cx = in.cx; mi.xf = in.xf; mi.cy = in.cy; mi.yf = in.yf;
      }
      vd fxv(0.), fyv0(0.), fyv1(0.);
      st.step(mi, jc >= jA, j >= jA, fxv, fyv0, fyv1);   // (the warm-up steps skip the sweeps that do not exist yet: no memory operation inside)
      // ---- KE flux + divergence damping at corner row jc (:1078-1198, :1372-1460) -----------------------------------------
      yv.push(in.v0);
      vd ke(0.);
      {
        const int jcc = jc < jA ? jA : jc;   // the row of the metric loads / the dropped store of the warm-up steps
        vd vb = dt5 * (shr1(in.vc) + in.vc);                               // :1129
        vd ub2 = dt5 * (uc_p + in.uc);                                     // :1186
        if constexpr (CS) {  // the interior of a cubed-sphere face (grid_type < 3): :1104-1106, :1168-1170
          const int jr = jcc > g.je + 1 ? g.je + 1 : jcc;
          const vd cosa = vload(g.cosa, (long)g.iB(ilo, jr), s.A);
          const vd rsina = vload(a.rsina, (long)(jr - g.js) * (g.nx + 1) + (ilo - g.is), s.F);
          const vd vs = shr1(in.vc) + in.vc, us = uc_p + in.uc;
          vb = dt5 * (vs - us * cosa) * rsina;
          ub2 = dt5 * (us - vs * cosa) * rsina;
        }
        const vd ub = yv.face(vb, UNI ? in.rdy : rdy_p, in.rdy);           // ytp_v :1134
        const vd kev = vb * ub;                                            // :1139
        const vd vb2 = ppm_faces_x_sw<SWC>(in.ukc, ub2, in.rdx);           // xtp_u :1191
        ke = 0.5 * (kev + ub2 * vb2);                                      // :1196
        const vd vc2 = (shl1(d_0) - d_0) * in.dgu;                         // :1392-1396
        const vd uc2m = (d_0 - d_m) * (UNI ? in.dgv : dgv_p);              // :1399-1403
        const vd uc2 = (in.dv - d_0) * in.dgv;
        vd lap = uc2m - uc2 + shr1(vc2) - vc2;                             // :1406-1424
        if (!g.stretched_grid) lap = lap * in.rac;
        vd dsave = d_0;                                                    // delpc = saved divergence (:1376-1381)
        if constexpr (UNI) {
          if (nord0) {  // v(jc-1), v(jc) are rows r-3, r-2 of the window of ytp_v; u(jc) was loaded for xtp_u
            const vd ptc = in.ukc * g.c_dyc;
            dsave = g.c_rarea_c * (yv.q1 * g.c_dxc - yv.q2 * g.c_dxc + shr1(ptc) - ptc);          // :1355-1366
            const vd damp = g.da_min_c * vmax(vd(d2_bg), vmin(vd(0.20), a.dddmp * vabs(dsave * a.dt)));
            ke = ke + damp * dsave;                                        // :1367-1369
          } else {
            ke = ke + (damp2 * d_0 + dd8 * lap);                           // :1455
          }
        } else {
          ke = ke + (damp2 * d_0 + dd8 * lap);
        }
        const bool on = dpc && jc >= jA && (jc <= jB || jc == g.je + 1) && jc >= oJ0 && jc <= oJ1;
        vstore_b(dpc ? dpc : uo, (long)g.iA(ilo, jcc), dsave, mOF, on);
      }
      // ---- D-grid wind update of row j (:1500-1509) ------------------------------------------------------------------------------
      {
        const bool on = j >= jA && j >= oJ0 && j <= oJ1;
        const vd ke0 = ke_p, ke1 = ke;
        const vd vtdx_j = u_j * dx_j, utdy_j = v_j * dy_j;
        vstore_b(uo, oUj, vtdx_j + ke0 - shl1(ke0) + fyv0 * yf_p, mO, on);
        vstore_b(vo, oVj, utdy_j + ke0 - ke1 - fxv * xf_j, mOF, on);
      }
      // ---- rotate ----------------------------------------------------------------------------------------------------------------------
      vtdx_n = vt1;
      if (r >= jA + 1) {  // the corner row loaded at this step was a real one (jA-1 or later)
        uc_p = in.uc;
        if constexpr (!UNI) { rdy_p = in.rdy; dgv_p = in.dgv; }
        d_m = d_0; d_0 = in.dv;
      }
      ke_p = ke;
      yf_p = yf_r;
      fyv1_last = fyv1;
    }
    // the north edge row of u: the last step (corner row je + 1, face je + 1) left ke, yfx and the face value
    if (jB == g.je && g.je >= oJ0 && g.je <= oJ1) {
      const long oU1 = (long)g.iU(ilo, g.je + 1);
      const vd dx_n = UNI ? vd(g.c_dx) : vload(g.dx, oU1, s.A);
      vstore(uo, oU1, vload(u, oU1, s.A) * dx_n + ke_p - shl1(ke_p) + fyv1_last * yf_p, s.lC0, s.lC1);
    }
  }

  FV3_D void run_general(int gid) const {
    int strip, seg, kk;
    md.decode(gid, strip, seg, kk);
    const int k = md.klist ? md.klist[kk] : kk;
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int rlast = jB + 3;
    const int lFx1 = (ilo + s.lC1 == g.ie) ? s.lC1 + 1 : s.lC1;
    const double *u = a.u + (size_t)k * g.nU(), *v = a.v + (size_t)k * g.nV();
    const double *uc = a.uc + (size_t)k * g.nV(), *vc = a.vc + (size_t)k * g.nU();
    const double *dv = a.divg_d + (size_t)k * g.nB();
    const double *crx = a.crx + (size_t)k * g.nCX(), *xfx = a.xfx + (size_t)k * g.nCX();
    const double *cry = a.cry + (size_t)k * g.nCY(), *yfx = a.yfx + (size_t)k * g.nCY();
    double *uo = a.u_out + (size_t)k * g.nU(), *vo = a.v_out + (size_t)k * g.nV();
    double *dpc = a.delpc ? a.delpc + (size_t)k * g.nA() : nullptr;
    const double dt5 = 0.5 * a.dt;
    // cubed-sphere hybrid: lanes / rows of the outputs this kernel owns (DswArgs::mask_w)
    const int mw = CS ? a.mask_w : 0;
    const int oC0 = (CS && mw) ? (s.lC0 > mw + 1 - ilo ? s.lC0 : mw + 1 - ilo) : s.lC0;
    const int oC1 = (CS && mw) ? (s.lC1 < g.npx - mw - 1 - ilo ? s.lC1 : g.npx - mw - 1 - ilo) : s.lC1;
    const int oF1 = (CS && mw) ? (lFx1 < g.npx - mw - 1 - ilo ? lFx1 : g.npx - mw - 1 - ilo) : lFx1;
    const int oJ0 = (CS && mw) ? mw + 1 : g.jsd, oJ1 = (CS && mw) ? g.npy - mw - 1 : g.jed + 1;
    const double d2_bg = a.lv.d2_divg[k];
    const double damp2 = g.da_min_c * dmax(d2_bg, dmin(0.20, a.dddmp * 0.));            // :1454 with vort = 0
    const double dd8 = g.stretched_grid ? g.da_min * ipow(a.d4_bg, 2) : ipow(g.da_min_c * a.d4_bg, 2);  // :1446-1450

    auto load_in = [&](int r) {
      In in;
      const long oU1 = (long)g.iU(ilo, r + 1), oV = (long)g.iV(ilo, r), oA = (long)g.iA(ilo, r);
      in.u1 = vload(u, oU1, s.A);
      in.v0 = vload(v, oV, s.A);
      in.v1 = vload(v, oV + 1, s.A);  // V kind has the extra column ied+1
      in.f0 = vload(g.f0, oA, s.A);
      if constexpr (UNI) {
        in.dx1 = vd(g.c_dx);  in.dy0 = in.dy1 = vd(g.c_dy);
        in.ra = vd(g.c_rarea);  in.ar = vd(g.c_area);
      } else {
        in.dx1 = vload(g.dx, oU1, s.A);
        in.dy0 = vload(g.dy, oV, s.A);  in.dy1 = vload(g.dy, oV + 1, s.A);
        in.ra = vload(g.rarea, oA, s.A);
        in.ar = vload(g.area, oA, s.A);
      }
      const long oCX = (long)g.iCX(ilo, r);
      in.cx = vload(crx, oCX, s.F);
      in.xf = vload(xfx, oCX, s.F);
      const int jf = (r - 2 < jA) ? jA : r - 2;
      const long oCY = (long)g.iCY(ilo, jf);
      in.cy = vload(cry, oCY, s.A);
      in.yf = vload(yfx, oCY, s.A);
      // corner row jc = r-2; before the segment's first corner row the row below it is loaded, whose values become
      // the "previous row" carries of the first one
      const int jc = (r - 2 < jA - 1) ? jA - 1 : r - 2;
      const long oUc = (long)g.iU(ilo, jc), oVc = (long)g.iV(ilo, jc), oBc = (long)g.iB(ilo, jc);
      in.vc = vload(vc, oUc, s.A);
      in.uc = vload(uc, oVc, s.A);
      in.ukc = vload(u, oUc, s.A);
      in.dv = vload(dv, (long)g.iB(ilo, jc + 1), s.A);
      if constexpr (UNI) {
        in.rdy = vd(g.c_rdy);  in.rdx = vd(g.c_rdx);
        in.dgu = vd(g.c_divg_u);  in.dgv = vd(g.c_divg_v);  in.rac = vd(g.c_rarea_c);
      } else {
        in.rdy = vload(g.rdy, oVc, s.A);
        in.rdx = vload(g.rdx, oUc, s.A);
        in.dgu = vload(g.divg_u, oUc, s.A);
        in.dgv = vload(g.divg_v, oVc, s.A);
        in.rac = vload(g.rarea_c, oBc, s.A);
      }
      return in;
    };

    Tp2dState<HORD, UNI> st;
    st.init();
    PpmYsw<SWC> yv;
    yv.init();
    vd vtdx_n(0.);                                  // u(r)*dx(r): the "vt1" of the previous step
    vd uc_p(0.), rdy_p(0.), dgv_p(0.), d_m(0.), d_0(0.);  // corner row jc-1 carries; divg_d(jc-1), divg_d(jc)
    vd ke_p(0.), yf_p(0.);
    {  // u(jA-3)*dx(jA-3) and divg_d(jA-1) for the first steps
      const long oU = (long)g.iU(ilo, jA - 3);
      vtdx_n = vload(u, oU, s.A) * (UNI ? vd(g.c_dx) : vload(g.dx, oU, s.A));
      d_0 = vload(dv, (long)g.iB(ilo, jA - 1), s.A);
    }
    In nxt = load_in(jA - 3);
    for (int r = jA - 3; r <= rlast; r++) {
      const In in = nxt;
      const int j = r - 3, jc = r - 2;
      // u, dx, v, dy, xfx of row j for the wind update at the end of this step: re-read (L1/L2 hits, issued first so
      // they are back long before they are used) rather than carried in registers for three steps
      const int jw = j >= jA ? j : jA;
      const long oUj = (long)g.iU(ilo, jw), oVj = (long)g.iV(ilo, jw);
      const vd u_j = vload(u, oUj, s.A), dx_j = UNI ? vd(g.c_dx) : vload(g.dx, oUj, s.A);
      const vd v_j = vload(v, oVj, s.A), dy_j = UNI ? vd(g.c_dy) : vload(g.dy, oVj, s.A);
      const vd xf_j = vload(xfx, (long)g.iCX(ilo, jw), s.F);
      nxt = load_in(r < rlast ? r + 1 : rlast);
      // ---- absolute vorticity of row r (sw_core.F90:1231-1247, :1476-1495) -> fv_tp_2d march ---------------------------
      const vd vt0 = vtdx_n, vt1 = in.u1 * in.dx1, ut0 = in.v0 * in.dy0, ut1 = in.v1 * in.dy1;
      MarchIn mi;
      mi.qn = in.ra * (vt0 - vt1 - ut0 + ut1) + in.f0;
      mi.ar = in.ar; mi.cx = in.cx; mi.xf = in.xf; mi.cy = in.cy; mi.yf = in.yf;
      vd fxv, fyv0, fyv1;
      st.step(mi, jc >= jA, j >= jA, fxv, fyv0, fyv1);
      // ---- KE flux + divergence damping at corner row jc (:1078-1198, :1372-1460) -----------------------------------------
      yv.push(in.v0);
      vd ke(0.);
      if (jc >= jA) {
        vd vb = dt5 * (shr1(in.vc) + in.vc);                               // :1129
        vd ub2 = dt5 * (uc_p + in.uc);                                     // :1186
        if constexpr (CS) {  // the interior of a cubed-sphere face (grid_type < 3): :1104-1106, :1168-1170
          const int jr = jc > g.je + 1 ? g.je + 1 : jc;
          const vd cosa = vload(g.cosa, (long)g.iB(ilo, jr), s.A);
          const vd rsina = vload(a.rsina, (long)(jr - g.js) * (g.nx + 1) + (ilo - g.is), s.F);
          const vd vs = shr1(in.vc) + in.vc, us = uc_p + in.uc;
          vb = dt5 * (vs - us * cosa) * rsina;
          ub2 = dt5 * (us - vs * cosa) * rsina;
        }
        const vd ub = yv.face(vb, UNI ? in.rdy : rdy_p, in.rdy);           // ytp_v :1134
        const vd kev = vb * ub;                                            // :1139
        const vd vb2 = ppm_faces_x_sw<SWC>(in.ukc, ub2, in.rdx);           // xtp_u :1191
        ke = 0.5 * (kev + ub2 * vb2);                                      // :1196
        const vd vc2 = (shl1(d_0) - d_0) * in.dgu;                         // :1392-1396
        const vd uc2m = (d_0 - d_m) * (UNI ? in.dgv : dgv_p);              // :1399-1403
        const vd uc2 = (in.dv - d_0) * in.dgv;
        vd lap = uc2m - uc2 + shr1(vc2) - vc2;                             // :1406-1424
        if (!g.stretched_grid) lap = lap * in.rac;
        ke = ke + (damp2 * d_0 + dd8 * lap);                               // :1455
        if (dpc && (jc <= jB || jc == g.je + 1) && jc >= oJ0 && jc <= oJ1)
          vstore(dpc, (long)g.iA(ilo, jc), d_0, oC0, oF1);                 // delpc = saved divergence (:1376-1381)
      }
      // ---- D-grid wind update of row j (:1500-1509) ------------------------------------------------------------------------------
      if (j >= jA && j >= oJ0 && j <= oJ1) {
        const vd ke0 = ke_p, ke1 = ke;
        const vd vtdx_j = u_j * dx_j, utdy_j = v_j * dy_j;
        vstore(uo, oUj, vtdx_j + ke0 - shl1(ke0) + fyv0 * yf_p, oC0, oC1);
        vstore(vo, oVj, utdy_j + ke0 - ke1 - fxv * xf_j, oC0, oF1);
        if (j == g.je) {  // the north edge row of u
          const long oU1 = (long)g.iU(ilo, j + 1);
          const vd dx_n = UNI ? vd(g.c_dx) : vload(g.dx, oU1, s.A);
          vstore(uo, oU1, vload(u, oU1, s.A) * dx_n + ke1 - shl1(ke1) + fyv1 * in.yf, s.lC0, s.lC1);
        }
      }
      // ---- rotate ----------------------------------------------------------------------------------------------------------------------
      vtdx_n = vt1;
      if (r >= jA + 1) {  // the corner row loaded at this step was a real one (jA-1 or later)
        uc_p = in.uc;
        if constexpr (!UNI) { rdy_p = in.rdy; dgv_p = in.dgv; }
        d_m = d_0; d_0 = in.dv;
      }
      ke_p = ke;
      if (jc >= jA) yf_p = in.yf;
    }
  }
};

}  // namespace fv3

// =====================================================================================================
// TracerMarchFused: one sub-cycle of tracer_2d (fv_tracer2d.F90:471-541, trdm <= 1e-4) for NT tracers of one level per
// wavefront.  TracerMarch (dsw_march.h) runs one (tracer, level) per wavefront and re-reads dp1, mfx, mfy, crx, cry, xfx,
// yfx, area, rarea for every tracer: 92 B per cell and tracer measured against 16 + 64/nq algorithmic.  Here the NT marches
// share those rows (and the reciprocals of ra_x, ra_y, dp2), so a tracer costs its own row in and out.
namespace fv3 {

template <int HORD, int NT>
struct TracerMarchFused {
  Grid g;
  MarchDims md;
  int npz, nq, it, nsplt, ngrp;  // ngrp = ceil(nq / NT) tracer groups
  const int *ksplt;              // device, npz
  const double *q, *dp1, *mfx, *mfy, *cx, *cy, *xfx, *yfx;
  double *q_out, *dp1_out;

  template <int NL>
  struct In {
    vd q[NL];              // row r of each tracer
    vd ar, cx, xf;         // row r: area, crx, xfx
    vd cy, yf;             // face r-2
    vd cxj, d1, ra, mx, my0, my1;  // row j = r-3: crx, dp1, rarea, mfx, mfy(j), mfy(j+1)
  };

  FV3_D void operator()(int gid) const {
    int strip, seg, kq;
    md.decode(gid, strip, seg, kq);
    const int k = kq % npz, grp = kq / npz;
    // the nq tracers are dealt evenly to the ngrp groups (4 tracers: 2 + 2, not 3 + 1 -- a one-tracer march at the
    // occupancy of this kernel is slow); a group runs the march instantiated for its size
    const int base = nq / ngrp, rem = nq % ngrp;
    const int nl = base + (grp < rem ? 1 : 0), iq0 = grp * base + (grp < rem ? grp : rem);
    if (nl == NT) { run<NT>(strip, seg, k, grp, iq0); return; }
    if constexpr (NT > 1) if (nl == 1) { run<1>(strip, seg, k, grp, iq0); return; }
    if constexpr (NT > 2) if (nl == 2) { run<2>(strip, seg, k, grp, iq0); return; }
    if constexpr (NT > 3) if (nl == 3) { run<3>(strip, seg, k, grp, iq0); return; }
  }

  template <int NL>
  FV3_D void run(int strip, int seg, int k, int grp, int iq0) const {
    const StripGeom s = make_strip(g, strip);
    const int ilo = s.ilo;
    const int jA = g.js + seg * md.tj;
    const int jB = (jA + md.tj - 1 < g.je) ? jA + md.tj - 1 : g.je;
    const int rlast = jB + 3;
    const size_t oA = (size_t)k * g.nA(), oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    auto qoff = [&](int t) { return ((size_t)(iq0 + t) * npz + k) * g.nA(); };
    const bool write_dp = (grp == ngrp - 1) && it != nsplt;   // as TracerMarch: the last tracer's wavefront carries dp1
    if (it > ksplt[k]) {  // the level is finished: carry q (and dp1) over to the output buffers
      for (int j = jA; j <= jB; j++) {
        const long iA = (long)g.iA(ilo, j);
        for (int t = 0; t < NL; t++) vstore(q_out + qoff(t), iA, vload(q + qoff(t), iA, s.A), s.lC0, s.lC1);
        if (write_dp) vstore(dp1_out + oA, iA, vload(dp1 + oA, iA, s.A), s.lC0, s.lC1);
      }
      return;
    }
    auto load_in = [&](int r) {
      In<NL> in;
      const long iA = (long)g.iA(ilo, r), iCX = (long)g.iCX(ilo, r);
      for (int t = 0; t < NL; t++) in.q[t] = vload(q + qoff(t), iA, s.A);
      in.ar = g.geom == 2 ? vd(g.c_area) : vload(g.area, iA, s.A);
      in.cx = vload(cx + oCX, iCX, s.F);
      in.xf = vload(xfx + oCX, iCX, s.F);
      const int jf = (r - 2 < jA) ? jA : r - 2, j = (r - 3 < jA) ? jA : r - 3;
      const long iCY = (long)g.iCY(ilo, jf);
      in.cy = vload(cy + oCY, iCY, s.A);
      in.yf = vload(yfx + oCY, iCY, s.A);
      const long iAj = (long)g.iA(ilo, j);
      in.cxj = vload(cx + oCX, (long)g.iCX(ilo, j), s.F);
      in.d1 = vload(dp1 + oA, iAj, s.A);
      in.ra = g.geom == 2 ? vd(g.c_rarea) : vload(g.rarea, iAj, s.A);
      in.mx = vload(mfx + (size_t)k * g.nFX(), (long)g.iFX(ilo, j), s.F);
      in.my0 = vload(mfy + (size_t)k * g.nFY(), (long)g.iFY(ilo, j), s.C);
      in.my1 = vload(mfy + (size_t)k * g.nFY(), (long)g.iFY(ilo, j + 1), s.C);
      return in;
    };

    Tp2dField<HORD> fd[NL];
    for (int t = 0; t < NL; t++) fd[t].init();
    vd ar_1(1.), ar_2(1.), ar_3(1.);  // area of rows r-1 .. r-3
    vd yf_prev(0.);
    In<NL> nxt = load_in(jA - 3);
#if FV3_BF
    // the row step without a branch around a store (DswTransportFused::run_bf): the warm-up steps form the update of row jA from the
    // clamped rows they loaded and drop it (`on` = false: a buffer store with no records)
    // (tools/tz_ab.py at C384 L127, same arrays: 4 tracers 0.86 -> 0.80 ms, 33 tracers 6.58 -> 6.43 ms per sub-cycle: the kernel is bound by
    // the f64 issue rate of fv_tp_2d itself -- 96 G tracer-cells/s -- not by its memory waits)
    const vm mC = make_mask(s.lC0, s.lC1);
    vdrain_loads();
#endif
    for (int r = jA - 3; r <= rlast; r++) {
      const In<NL> in = nxt;
      nxt = load_in(r < rlast ? r + 1 : rlast);
      const int j = r - 3;
      const bool have_face = r - 2 >= jA, have_row = j >= jA;
      Tp2dShared sh;
      sh.cx = in.cx; sh.xf = in.xf; sh.cy = in.cy; sh.yf = in.yf;
      sh.ar = in.ar;
      sh.rax = in.ar + sh.xf - shl1(sh.xf);
      sh.arj = ar_3; sh.cxj = in.cxj;
      sh.ray = sh.arj + yf_prev - sh.yf;
      sh.rrax = vrecip(sh.rax);
      sh.rray = vrecip(sh.ray);
      ar_3 = ar_2; ar_2 = ar_1; ar_1 = in.ar;
      vd fx[NL], fy0[NL], fy1[NL];
      for (int t = 0; t < NL; t++) fd[t].step(in.q[t], sh, have_face, have_row, fx[t], fy0[t], fy1[t]);
#if FV3_BF
      {
        const bool on = have_face && have_row;
        const long iA = (long)g.iA(ilo, j < jA ? jA : j);
        const vd dp2 = in.d1 + (in.mx - shl1(in.mx) + in.my0 - in.my1) * in.ra;                 // :517-522
        const vd rdp2 = vrecip(dp2);
        for (int t = 0; t < NL; t++) {
          const vd gx = fx[t] * in.mx, gy0 = fy0[t] * in.my0, gy1 = fy1[t] * in.my1;
          const vd qn = vdiv_r(fd[t].ya.row_m3() * in.d1 + (gx - shl1(gx) + gy0 - gy1) * in.ra, dp2, rdp2);  // :523-531
          vstore_b_nt(q_out + qoff(t), iA, qn, mC, on);
        }
        vstore_b_nt(dp1_out + oA, iA, dp2, mC, on && write_dp);
      }
#else
      if (have_face && have_row) {
        const long iA = (long)g.iA(ilo, j);
        const vd dp2 = in.d1 + (in.mx - shl1(in.mx) + in.my0 - in.my1) * in.ra;                 // :517-522
        const vd rdp2 = vrecip(dp2);
        for (int t = 0; t < NL; t++) {
          const vd gx = fx[t] * in.mx, gy0 = fy0[t] * in.my0, gy1 = fy1[t] * in.my1;
          const vd qn = vdiv_r(fd[t].ya.row_m3() * in.d1 + (gx - shl1(gx) + gy0 - gy1) * in.ra, dp2, rdp2);  // :523-531
          vstore(q_out + qoff(t), iA, qn, s.lC0, s.lC1);
        }
        if (write_dp) vstore(dp1_out + oA, iA, dp2, s.lC0, s.lC1);
      }
#endif
      if (have_face) yf_prev = sh.yf;
    }
  }
};

}  // namespace fv3
