// ppm_march.h -- 1-D PPM in "per-cell then per-face" form for the wave-marching kernels.
//
// Reference: model/tp_core.F90 xppm :324-712 / yppm :715-1152 and model/sw_core.F90 xtp_u :2154-2521 /
// ytp_v :2524-2998 (grid_type >= 3 branches).  The reference builds al / bl / br for a whole row and
// then picks the upwind cell per face; that is exactly the shape kept here: every per-cell quantity
// (dm, al, bl, br, smoothness flag) is evaluated ONCE per cell, a face then combines its two
// neighbouring cells.  Along i the neighbours come from wavefront shifts, along j from a register
// window (PpmY).  Expression order follows the reference so that, with FMA contraction off, results
// are bit-identical to the oracle.
#pragma once

#include "spmd.h"

namespace fv3 {

// what a face needs from each of its two cells
struct PCell {
  vd q, bl, br;
  vb smt;  // smoothness flag (unlimited family only)
};

// monotone slope (tp_core.F90:570-574 == sw_core.F90:2383-2387)
FV3_D vd ppm_dm_v(const vd &qm, const vd &q0, const vd &qp) {
  const vd xt = 0.25 * (qp - qm);
  return vsign(vmin3(vabs(xt), vmax3(qm, q0, qp) - q0, q0 - vmin3(qm, q0, qp)), xt);
}

// al for the monotone family: edge between cells (i-1, i)  (tp_core.F90:576)
FV3_D vd ppm_al_mono(const vd &qm1, const vd &q0, const vd &dmm, const vd &dm0) {
  constexpr double r3 = 1. / 3.;
  return 0.5 * (qm1 + q0) + r3 * (dmm - dm0);
}
// al for the unlimited family (tp_core.F90:365-369): edge between cells (i-1, i)
template <int ORD>
FV3_D vd ppm_al_unlim(const vd &qm2, const vd &qm1, const vd &q0, const vd &qp1) {
  constexpr double p1 = 7. / 12., p2 = -1. / 12.;
  vd al = p1 * (qm1 + q0) + p2 * (qm2 + qp1);
  if (ORD < 0) al = vmax(vd(0.), al);
  return al;
}

// ---- tp_core flavour: bl, br (and flag) of one cell ---------------------------------------------
// ORD in {8, 10}: needs al0 = al(i), al1 = al(i+1), the three slopes and the 5-point window.
template <int ORD>
FV3_D PCell ppm_cell_mono(const vd &qm2, const vd &qm1, const vd &q0, const vd &qp1, const vd &qp2, const vd &dmm,
                          const vd &dm0, const vd &dmp, const vd &al0, const vd &al1) {
  PCell c;
  c.q = q0;
  if (ORD == 8) {  // tp_core.F90:579-584
    const vd xt = 2. * dm0;
    c.bl = -vsign(vmin(vabs(xt), vabs(al0 - q0)), xt);
    c.br = vsign(vmin(vabs(xt), vabs(al1 - q0)), xt);
  } else if (ORD == 11) {  // :604-610, ppm_fac = 1.5
    const vd xt = 1.5 * dm0;
    c.bl = -vsign(vmin(vabs(xt), vabs(al0 - q0)), xt);
    c.br = vsign(vmin(vabs(xt), vabs(al1 - q0)), xt);
  } else if (ORD == 12 || ORD == 9 || ORD == 7) {  // :611-633 / :634-641 with pert_ppm(iv = 0), :1219-1242 (13 runs as 9)
    constexpr double r12 = 1. / 12.;
    const vd bl0 = al0 - q0, br0 = al1 - q0;
    const vd a4 = -3. * (bl0 + br0), da1 = br0 - bl0;
    // lanes that fail the first test may divide by a4 = 0: their value is not selected
    const vb fix = (vabs(da1) < -a4) && (q0 + 0.25 / a4 * (da1 * da1) + a4 * r12 < 0.);
    const vb both = (ORD == 9) ? ((br0 > 0.) && (bl0 > 0.)) : (br0 * bl0 > 0.);
    const vb up = da1 > 0.;
    vd bl = vsel(fix, vsel(both, vd(0.), vsel(up, bl0, -2. * br0)), bl0);
    vd br = vsel(fix, vsel(both, vd(0.), vsel(up, -2. * bl0, br0)), br0);
    if (ORD == 9) {
      const vb empty = q0 <= 0.;
      bl = vsel(empty, vd(0.), bl);
      br = vsel(empty, vd(0.), br);
    }
    c.bl = bl;
    c.br = br;
    if (ORD == 7) c.smt = bl * br < 0.;  // :685-689: iord = 7 takes the flux form of the unlimited family on these cells
  } else {  // ORD == 10, :585-603
    constexpr double near_zero = 1.E-25;
    const vd bl0 = al0 - q0, br0 = al1 - q0;
    const vb flat = vabs(dmm) + vabs(dm0) + vabs(dmp) < near_zero;
    const vb steep = vabs(3. * (bl0 + br0)) > vabs(bl0 - br0);
    // 0.75*(2.*d) of the reference (tp_core.F90:587,597) is 1.5*d bit for bit: 2.*d is exact, so both round 1.5*d once
    const vd pmp_2 = 2. * (q0 - qm1);
    const vd lac_2 = pmp_2 - 1.5 * (qm1 - qm2);
    const vd brl = vmin(vmax3(vd(0.), pmp_2, lac_2), vmax(br0, vmin3(vd(0.), pmp_2, lac_2)));
    const vd pmp_1 = -(2. * (qp1 - q0));
    const vd lac_1 = pmp_1 + 1.5 * (qp2 - qp1);
    const vd bll = vmin(vmax3(vd(0.), pmp_1, lac_1), vmax(bl0, vmin3(vd(0.), pmp_1, lac_1)));
    c.bl = vsel(flat, vd(0.), vsel(steep, bll, bl0));
    c.br = vsel(flat, vd(0.), vsel(steep, brl, br0));
  }
  return c;
}

// ORD in {5, -5, 6}: al0 = al(i), al1 = al(i+1)  (tp_core.F90:371-397, :499-524)
template <int ORD>
FV3_D PCell ppm_cell_unlim(const vd &q0, const vd &al0, const vd &al1) {
  constexpr double r12 = 1. / 12.;
  PCell c;
  c.q = q0;
  vd bl = al0 - q0, br = al1 - q0;
  const vd b0 = bl + br;
  if (ORD == 6)
    c.smt = 3. * vabs(b0) < vabs(bl - br);
  else
    c.smt = bl * br < 0.;
  if (ORD == -5) {  // positive-definite adjustment, :499-524
    const vd da1 = br - bl, a4 = -3. * b0;
    const vb act = (vabs(da1) < -a4) && (q0 + 0.25 / a4 * (da1 * da1) + a4 * r12 < 0.);
    const vb up = da1 > 0.;
    // !smt: flat; smt & da1>0: br = -2 bl; else bl = -2 br
    const vd bl_n = vsel(!c.smt, vd(0.), vsel(up, bl, -2. * br));
    const vd br_n = vsel(!c.smt, vd(0.), vsel(up, -2. * bl, br));
    bl = vsel(act, bl_n, bl);
    br = vsel(act, br_n, br);
  }
  c.bl = bl;
  c.br = br;
  return c;
}

// ---- face value between cell m (left / below) and cell p; c = Courant number at the face -----------
// tp_core.F90:549-558 (unlimited) and :701-707 (monotone).  (1+c) == 1-|c| and bl + c*b0 == bl - |c|*b0
// bit for bit when c <= 0, so one formula serves both signs.
template <int ORD>
FV3_D vd ppm_face_v(const PCell &m, const PCell &p, const vd &c) {
  const vb pos = c > 0.;
  const vd s = vabs(c);
  const vd qu = vsel(pos, m.q, p.q);
  const vd bl = vsel(pos, m.bl, p.bl), br = vsel(pos, m.br, p.br);
  const vd x = vsel(pos, br, bl);
  const vd fx1 = (1. - s) * (x - s * (bl + br));
  if (ORD >= 8) return qu + fx1;
  return vsel(m.smt || p.smt, qu + fx1, qu);
}

// ---- sw_core flavour (xtp_u / ytp_v, "other grids" branches) -------------------------------------------
// monotone ORD >= 8: sw_core.F90:2492-2516 / :2973-2996
FV3_D PCell ppm_cell_sw_mono(const vd &qm2, const vd &qm1, const vd &q0, const vd &qp1, const vd &qp2, const vd &al0,
                             const vd &al1) {
  PCell c;
  c.q = q0;
  vd pmp = -2. * (qp1 - q0);
  vd lac = pmp + 1.5 * (qp2 - qp1);
  c.bl = vmin(vmax3(vd(0.), pmp, lac), vmax(al0 - q0, vmin3(vd(0.), pmp, lac)));
  pmp = 2. * (q0 - qm1);
  lac = pmp - 1.5 * (qm1 - qm2);
  c.br = vmin(vmax3(vd(0.), pmp, lac), vmax(al1 - q0, vmin3(vd(0.), pmp, lac)));
  return c;
}
// the cubed-sphere branch (grid_type < 3) away from the face edges, iord = 8, 10, 11 (sw_core.F90:2381-2436; iord = 9 is the
// form above): classes 108, 110, 111 of the marching kernels (DswMomentumFused on a cubed-sphere face)
template <int SWC>
FV3_D PCell ppm_cell_sw_cs(const vd &qm2, const vd &qm1, const vd &q0, const vd &qp1, const vd &qp2, const vd &dmm, const vd &dm0,
                           const vd &dmp, const vd &al0, const vd &al1) {
  PCell c;
  c.q = q0;
  if (SWC == 108) {
    const vd xt = 2. * dm0;
    c.bl = -vsign(vmin(vabs(xt), vabs(al0 - q0)), xt);
    c.br = vsign(vmin(vabs(xt), vabs(al1 - q0)), xt);
  } else if (SWC == 110) {
    const vd bl = al0 - q0, br = al1 - q0;
    const vb flat = vabs(dm0) < 1.E-9;
    const vb two_dx = vabs(dmm) + vabs(dmp) < 1.E-9;
    const vb limit = vabs(3. * (bl + br)) > vabs(bl - br);
    const vd pmp_1 = -2. * (qp1 - q0), lac_1 = pmp_1 + 1.5 * (qp2 - qp1);
    const vd pmp_2 = 2. * (q0 - qm1), lac_2 = pmp_2 - 1.5 * (qm1 - qm2);
    const vd bl_l = vmin(vmax3(vd(0.), pmp_1, lac_1), vmax(bl, vmin3(vd(0.), pmp_1, lac_1)));
    const vd br_l = vmin(vmax3(vd(0.), pmp_2, lac_2), vmax(br, vmin3(vd(0.), pmp_2, lac_2)));
    c.bl = vsel(flat, vsel(two_dx, vd(0.), bl), vsel(limit, bl_l, bl));
    c.br = vsel(flat, vsel(two_dx, vd(0.), br), vsel(limit, br_l, br));
  } else {
    c.bl = al0 - q0;
    c.br = al1 - q0;
  }
  return c;
}
// unlimited ORD in {5, 6, 7}: sw_core.F90:2190-2243, 2337-2374
template <int ORD>
FV3_D PCell ppm_cell_sw_unlim(const vd &q0, const vd &al0, const vd &al1) {
  PCell c;
  c.q = q0;
  c.bl = al0 - q0;
  c.br = al1 - q0;
  const vd b0 = c.bl + c.br;
  if (ORD == 5)
    c.smt = c.bl * c.br < 0.;
  else
    c.smt = 3. * vabs(b0) < vabs(c.bl - c.br);
  return c;
}
// face: c = advective displacement at the face, rdm / rdp = 1/dx (1/dy) of the two cells
template <int ORD>
FV3_D vd ppm_face_sw_v(const PCell &m, const PCell &p, const vd &c, const vd &rdm, const vd &rdp) {
  const vb pos = c > 0.;
  const vd cfl = vsel(pos, c * rdm, c * rdp);
  return ppm_face_v<ORD>(m, p, cfl);
}

// =====================================================================================================
// x direction: lanes hold consecutive cells of one row.  Returns the per-cell data; the face between
// lanes (l-1, l) is then ppm_face_v(shift_cell(c), c, courant).  Valid cells: lanes 2..61.
template <int ORD>
FV3_D PCell ppm_cells_x(const vd &q) {
  const vd qm1 = shr1(q), qp1 = shl1(q);
  const vd qm2 = shr1(qm1), qp2 = shl1(qp1);
  if (ORD >= 7) {
    const vd dm0 = ppm_dm_v(qm1, q, qp1);
    const vd dmm = shr1(dm0), dmp = shl1(dm0);
    const vd al0 = ppm_al_mono(qm1, q, dmm, dm0);
    const vd al1 = shl1(al0);
    return ppm_cell_mono<ORD>(qm2, qm1, q, qp1, qp2, dmm, dm0, dmp, al0, al1);
  } else {
    const vd al0 = ppm_al_unlim<ORD>(qm2, qm1, q, qp1);
    const vd al1 = shl1(al0);
    return ppm_cell_unlim<ORD>(q, al0, al1);
  }
}

FV3_D PCell shift_cell_r(const PCell &c, bool with_flag) {
  PCell m;
  m.q = shr1(c.q);
  m.bl = shr1(c.bl);
  m.br = shr1(c.br);
#ifdef FV3_HOST_EMU
  m.smt.v[0] = false;
  for (int l = 1; l < kW; l++) m.smt.v[l] = c.smt.v[l - 1];
  (void)with_flag;
#else
  m.smt = with_flag ? (__builtin_amdgcn_update_dpp(0, (int)c.smt, 0x138, 0xf, 0xf, true) != 0) : false;
#endif
  return m;
}

// face values of one row: face l lies between lanes l-1 and l; valid faces: lanes 3..61.
// ppm_face_v with the left cell taken through the shift inside the selects (vsel_shr) and bl + br formed per cell before the select
// (the same two numbers added: bit for bit the value of ppm_face_v)
template <int ORD>
FV3_D vd ppm_faces_x(const vd &q, const vd &c) {
  const PCell p = ppm_cells_x<ORD>(q);
  const vb pos = c > 0.;
  const vd s = vabs(c);
  const vd b0 = p.bl + p.br;
  const vd qu = vsel_shr(pos, p.q, p.q);
  const vd x = vsel_shr(pos, p.br, p.bl);
  const vd b0u = vsel_shr(pos, b0, b0);
  const vd fx1 = (1. - s) * (x - s * b0u);
  if (ORD >= 8) return qu + fx1;
#ifdef FV3_HOST_EMU
  vb ms;
  ms.v[0] = false;
  for (int l = 1; l < kW; l++) ms.v[l] = p.smt.v[l - 1];
#else
  const vb ms = __builtin_amdgcn_update_dpp(0, (int)p.smt, 0x138, 0xf, 0xf, true) != 0;
#endif
  return vsel(ms || p.smt, qu + fx1, qu);
}

// =====================================================================================================
// y direction: a register window fed one row per step.  After push(q(r)) the newest complete cell is
// r-2 and face(c) returns the value at face r-2 (between rows r-3 and r-2).
template <int ORD>
struct PpmY {
  vd q0, q1, q2, q3, q4;  // rows r-4 .. r
  vd dm1, dm2, dm3;       // slopes of rows r-3, r-2, r-1        (monotone family)
  vd al2, al3;            // edges (r-3|r-2) and (r-2|r-1)
  PCell prev, cur;        // cells r-3 and r-2

  FV3_D void init() {
    q0 = q1 = q2 = q3 = q4 = vd(0.);
    dm1 = dm2 = dm3 = vd(0.);
    al2 = al3 = vd(0.);
    prev.q = prev.bl = prev.br = vd(0.);
    cur = prev;
#ifdef FV3_HOST_EMU
    for (int l = 0; l < kW; l++) prev.smt.v[l] = cur.smt.v[l] = false;
#else
    prev.smt = cur.smt = false;
#endif
  }
  FV3_D void push(const vd &qn) {
    q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = qn;
    prev = cur;
    al2 = al3;
    if (ORD >= 7) {
      dm1 = dm2; dm2 = dm3;
      dm3 = ppm_dm_v(q2, q3, q4);                 // slope of row r-1
      al3 = ppm_al_mono(q2, q3, dm2, dm3);        // edge (r-2 | r-1)
      cur = ppm_cell_mono<ORD>(q0, q1, q2, q3, q4, dm1, dm2, dm3, al2, al3);
    } else {
      al3 = ppm_al_unlim<ORD>(q1, q2, q3, q4);    // edge (r-2 | r-1): rows r-3 .. r
      cur = ppm_cell_unlim<ORD>(q2, al2, al3);
    }
  }
  FV3_D vd face(const vd &c) const { return ppm_face_v<ORD>(prev, cur, c); }
  FV3_D const vd &row_m3() const { return q1; }  // q(r-3)
};

// =====================================================================================================
// sw_core flavour (xtp_u / ytp_v).  SWC = scheme class: 5 (iord 5), 6 (iord 6, 7), 8 (iord >= 8).
constexpr int sw_class(int iord) { return iord >= 8 ? 8 : (iord == 5 ? 5 : 6); }
// the classes of the cubed-sphere branch away from the face edges (iord = 9 is the "other grids" form)
constexpr int sw_class_cubed(int iord) { return iord == 8 ? 108 : (iord == 10 ? 110 : (iord == 11 ? 111 : sw_class(iord))); }

template <int SWC>
FV3_D PCell ppm_cells_x_sw(const vd &q) {
  const vd qm1 = shr1(q), qp1 = shl1(q);
  const vd qm2 = shr1(qm1), qp2 = shl1(qp1);
  if (SWC >= 8) {
    const vd dm0 = ppm_dm_v(qm1, q, qp1);
    const vd al0 = ppm_al_mono(qm1, q, shr1(dm0), dm0);
    if (SWC > 100) return ppm_cell_sw_cs<SWC>(qm2, qm1, q, qp1, qp2, shr1(dm0), dm0, shl1(dm0), al0, shl1(al0));
    return ppm_cell_sw_mono(qm2, qm1, q, qp1, qp2, al0, shl1(al0));
  } else {
    const vd al0 = ppm_al_unlim<5>(qm2, qm1, q, qp1);
    return ppm_cell_sw_unlim<SWC>(q, al0, shl1(al0));
  }
}
// face l between lanes l-1 and l; c = advective displacement at the face, rd = 1/dx of the cells
template <int SWC>
FV3_D vd ppm_faces_x_sw(const vd &q, const vd &c, const vd &rd) {
  const PCell p = ppm_cells_x_sw<SWC>(q);
  const PCell m = shift_cell_r(p, SWC < 8);
  return ppm_face_sw_v<(SWC >= 8 ? 8 : 5)>(m, p, c, shr1(rd), rd);
}

template <int SWC>
struct PpmYsw {
  vd q0, q1, q2, q3, q4;
  vd dm1, dm2, dm3;
  vd al2, al3;
  PCell prev, cur;
  FV3_D void init() {
    q0 = q1 = q2 = q3 = q4 = vd(0.);
    dm1 = dm2 = dm3 = vd(0.);
    al2 = al3 = vd(0.);
    prev.q = prev.bl = prev.br = vd(0.);
    cur = prev;
#ifdef FV3_HOST_EMU
    for (int l = 0; l < kW; l++) prev.smt.v[l] = cur.smt.v[l] = false;
#else
    prev.smt = cur.smt = false;
#endif
  }
  FV3_D void push(const vd &qn) {
    q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = qn;
    prev = cur;
    al2 = al3;
    if (SWC >= 8) {
      if (SWC > 100) dm1 = dm2;
      dm2 = dm3;
      dm3 = ppm_dm_v(q2, q3, q4);
      al3 = ppm_al_mono(q2, q3, dm2, dm3);
      if (SWC > 100)
        cur = ppm_cell_sw_cs<SWC>(q0, q1, q2, q3, q4, dm1, dm2, dm3, al2, al3);
      else
        cur = ppm_cell_sw_mono(q0, q1, q2, q3, q4, al2, al3);
    } else {
      al3 = ppm_al_unlim<5>(q1, q2, q3, q4);
      cur = ppm_cell_sw_unlim<SWC>(q2, al2, al3);
    }
  }
  // face r-2 (between rows r-3 and r-2); rdm / rdp = 1/dy of those two rows
  FV3_D vd face(const vd &c, const vd &rdm, const vd &rdp) const {
    return ppm_face_sw_v<(SWC >= 8 ? 8 : 5)>(prev, cur, c, rdm, rdp);
  }
};

// =====================================================================================================
// fv3_ppm_line: ONE line through the 1-D operators of the marching kernels -- the unit-test surface for the reference-held vectors of
// xppm / yppm (tests/golden/ppm1d_golden.npz), iord 10 among them: inside fv_tp_2d the inner sweep of hord 10 is ord 8
// (tp_core.F90:136-141), so those vectors cannot pass through it unchanged.  h: the line with its 3 halo cells on either side
// (n + 6 values, cell i at h[i + 2]), c: the n + 1 Courant numbers, flux: the n + 1 face values.
//   along == 0: the lanes hold the line (n + 6 <= 64): ppm_faces_x, the x sweeps of the marching kernels
//   along == 1: the register window PpmY fed cell after cell (every lane the same line): their y sweeps
template <int ORD>
struct PpmLineMarch {
  const double *h, *c;
  double *flux;
  int n, along;
  FV3_D void operator()(int) const {
    if (along == 0) {
      const vl li = make_lanes(0, n + 5), lf = make_lanes(3, n + 3);
      const vd q = vload(h, 0, li);
      const vd cf = vload(c, -3, lf);                       // face f = lane - 2 sits between lanes (l - 1, l): c[f - 1] = c[l - 3]
      const vd f = ppm_faces_x<ORD>(q, cf);
      vstore(flux, -3, f, 3, n + 3);
    } else {
      const vl l0 = make_lanes(0, 0);
      PpmY<ORD> w;
      w.init();
      for (int r = 0; r < n + 6; r++) {
        w.push(vload(h, r, l0));
        if (r >= 5) vstore(flux, r - 5, w.face(vload(c, r - 5, l0)), 0, 0);   // face f = r - 4 (1-based) between cells f - 1 and f
      }
    }
  }
};

}  // namespace fv3
