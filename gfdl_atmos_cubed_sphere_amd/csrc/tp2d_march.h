// tp2d_march.h -- fv_tp_2d (model/tp_core.F90:85-241) as a wave-marching stencil.
//
// One wavefront owns a strip of 64 consecutive i-columns (lane 0 = column ilo) and a segment of rows
// [jA, jB]; it loads one row of q per step and keeps everything else in registers:
//
//   step r (r = jA-3 .. jB+3), after loading q(:, r):
//     fx2(r)  = xppm(q(:, r), crx(:, r), ord_in)              -- lanes talk through DPP shifts
//     q_j(r)  = (q*area + xfx*fx2 - xfx(i+1)*fx2(i+1)) / ra_x                       (:171-178)
//     push q(r) -> PpmY A,   push q_j(r) -> PpmY B
//     fy2(r-2) = A.face(cry(r-2))  (ord_in)    fy(r-2) = B.face(cry(r-2))  (ord_ou)  (:147, :180)
//     q_i(r-3) = (q*area + yfx(r-3)*fy2(r-3) - yfx(r-2)*fy2(r-2)) / ra_y             (:150-159)
//     fx(r-3)  = xppm(q_i(:, r-3), crx(:, r-3), ord_ou)                              (:161)
//     sink.row(j = r-3, 0.5*(fx + fx2)(j), 0.5*(fy + fy2)(j), 0.5*(fy + fy2)(j+1))
//
// so each of the four PPM sweeps is evaluated once per cell (plus a 6-row warm-up per segment and a
// 6-lane overlap per strip), nothing is staged through LDS and there are no barriers.
// Lane validity: q on all lanes it could be loaded for; x-faces on lanes 3..61, cells on lanes 3..60.
#pragma once

#include <type_traits>

#include <cstdlib>

#include "ppm_march.h"

namespace fv3 {

constexpr int kStripCells = kW - 6;  // 58 cells per strip (lanes 3..60)

struct StripGeom {
  int ilo;         // i index of lane 0  (= first cell of the strip - 3)
  int lA0, lA1;    // lanes whose column lies inside the halo'd array  [isd, ied]
  int lF0, lF1;    // lanes whose x-face exists                        [is, ie+1]
  int lC0, lC1;    // lanes of the cells this strip owns (subset of 3..60, inside [is, ie])
  vl A, F, C;      // the same three ranges as clamped lane indices for vload
};

FV3_D StripGeom make_strip(const Grid &g, int strip) {
  StripGeom s;
  const int ic0 = g.is + strip * kStripCells;
  s.ilo = ic0 - 3;
  auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
  s.lA0 = clampi(g.isd - s.ilo, 0, kW - 1);
  s.lA1 = clampi(g.ied - s.ilo, 0, kW - 1);
  s.lF0 = clampi(g.is - s.ilo, 0, kW - 1);
  s.lF1 = clampi(g.ie + 1 - s.ilo, 0, kW - 1);
  s.lC0 = 3;
  s.lC1 = clampi(g.ie - s.ilo, 0, kW - 4);
  s.A = make_lanes(s.lA0, s.lA1);
  s.F = make_lanes(s.lF0, s.lF1);
  s.C = make_lanes(s.lC0, s.lC1);
  return s;
}
inline int num_strips(const Grid &g) { return (g.nx + kStripCells - 1) / kStripCells; }

// wave index -> (strip, row segment, level)
struct MarchDims {
  int nstrips, nsegs, tj;
  const int *klist;  // level of the n-th marching slab (device), or null = identity
  int nk;            // number of level slots of the launch (set by nwaves)
  int k_fast;        // 1: consecutive wavefronts = consecutive levels of the same (strip, segment)
  // sub-box of the (strip, segment) grid this launch covers (default: all) -- lets a caller run the strips /
  // segments that do not touch the halo while the halo exchange is still in flight, and the frame afterwards
  int s0, ns, g0, ng;
  int frame;         // 1: the launch covers the frame of the whole grid around its interior box instead of a box
  // Whole rounds of the chip (balance_segments below): alt_nk of the level slots are cut into alt_ng segments of alt_tj rows instead
  // of nsegs segments of tj rows, so that the launch has a few wavefronts less than a whole number of rounds of the resident wavefronts
  // instead of a few more (a last round that fills 1 % of the chip costs a third of a round).  Whole-grid launches only.
  int alt_nk, alt_ng, alt_tj;
  FV3_HD void set_box(int s0_, int ns_, int g0_, int ng_) { s0 = s0_; ns = ns_; g0 = g0_; ng = ng_; frame = 0; }
  FV3_HD void set_frame() { s0 = 0; ns = nstrips; g0 = 0; ng = nsegs; frame = 1; }
  FV3_HD int ncells() const { return frame ? nstrips * nsegs - (nstrips - 2) * (nsegs - 2) : ns * ng; }
  // wavefronts of the level slots below kk: the alt_nk slots with one segment less are spread evenly over the nk slots (slot k is one of
  // them when floor((k + 1) alt_nk / nk) > floor(k alt_nk / nk)) -- the XCDs take contiguous ranges of wavefront indices (wave_index), and
  // the longer wavefronts must not all land on one of them (measured: all on XCD 0 = +10 % for the whole launch)
  FV3_HD int alt_before(int kk) const { return (int)(((long)kk * alt_nk) / nk); }
  FV3_HD int waves_before(int kk) const { return nstrips * (kk * nsegs - alt_before(kk) * (nsegs - alt_ng)); }
  FV3_HD int nwaves(int npz) {
    nk = npz;
    if (alt_nk > npz) alt_nk = npz;
    if (frame || ns != nstrips || ng != nsegs || k_fast) alt_nk = 0;
    if (alt_nk > 0) return waves_before(npz);
    return ncells() * npz;
  }
  // decode + the rows per segment of this wavefront
  FV3_HD void decode_tj(int gid, int &strip, int &seg, int &kk, int &tjw) const {
    tjw = tj;
    if (alt_nk > 0) {
      int lo = 0, hi = nk;   // the last slot whose first wavefront is <= gid
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (waves_before(mid) <= gid) lo = mid; else hi = mid;
      }
      kk = lo;
      const int r = gid - waves_before(kk);
      const bool alt = alt_before(kk + 1) > alt_before(kk);
      if (alt) tjw = alt_tj;
      strip = r % nstrips;
      seg = r / nstrips;
      return;
    }
    decode(gid, strip, seg, kk);
  }
  // t-th cell of the frame: south row, north row, then the west and east columns between them
  FV3_HD void frame_cell(int t, int &strip, int &seg) const {
    if (t < nstrips) { strip = t; seg = 0; return; }
    t -= nstrips;
    if (t < nstrips) { strip = t; seg = nsegs - 1; return; }
    t -= nstrips;
    if (t < nsegs - 2) { strip = 0; seg = 1 + t; return; }
    t -= nsegs - 2;
    strip = nstrips - 1;
    seg = 1 + t;
  }
  // k fastest: the workgroups that are in flight together (round-robin over the 8 XCDs) work on the same
  // (strip, segment) at different levels, so the 2-D metric rows they all read are served by each XCD's L2
  FV3_HD void decode(int gid, int &strip, int &seg, int &kk) const {
    if (frame) {
      const int nc = ncells();
      kk = gid / nc;
      frame_cell(gid % nc, strip, seg);
    } else if (k_fast) {
      kk = gid % nk;
      const int t = gid / nk;
      strip = s0 + t % ns;
      seg = g0 + t / ns;
    } else {
      strip = s0 + gid % ns;
      seg = g0 + (gid / ns) % ng;
      kk = gid / (ns * ng);
    }
  }
};
inline int march_k_fast() {
  static const int v = [] {
    const char *e = std::getenv("FV3_MI355X_K_FAST");
    return e ? std::atoi(e) : 0;  // measured: no gain on MI355X (metric rows are served by L2 / Infinity Cache either way)
  }();
  return v;
}
inline MarchDims make_march_dims(const Grid &g, int tj) {
  MarchDims d;
  d.tj = tj;
  d.klist = nullptr;
  d.nk = 0;
  d.k_fast = march_k_fast();
  d.nstrips = num_strips(g);
  d.nsegs = (g.ny + tj - 1) / tj;
  d.alt_nk = d.alt_ng = d.alt_tj = 0;
  d.set_box(0, d.nstrips, 0, d.nsegs);
  return d;
}

// Segments for a launch of nlev level slots over `rows` rows that does not end in a nearly empty round of the chip.  round_waves = the
// wavefronts resident at once (CUs x SIMDs x wavefronts per SIMD of the kernel), tj_conf = the configured rows per segment.
// The segments get equal lengths (S = ceil(rows / tj_conf) segments of ceil(rows / S) rows: no 4-row last segment); and when the launch
// would have a few wavefronts more than a whole number of rounds -- 127 levels x 7 strips x 7 segments = 6223 = 3 x 2048 + 79: measured
// +12 % for 1.6 % more work than 125 levels, tools/pair_ab2.py -- the first L level slots are cut into S - 1 segments, L chosen so that
// the total lands just under the whole rounds.  (Other segment counts were measured and lose: fewer, longer segments put the wavefronts
// of a SIMD in lockstep -- two rounds of 83- and 102-row wavefronts are 9 % slower than three of 61.)
inline void balance_segments(MarchDims &d, int nlev, int rows, int round_waves, int tj_conf) {
  d.alt_nk = 0;
  const int S = (rows + tj_conf - 1) / tj_conf;
  d.tj = (rows + S - 1) / S;
  d.nsegs = (rows + d.tj - 1) / d.tj;
  d.set_box(0, d.nstrips, 0, d.nsegs);
  if (nlev <= 0 || round_waves <= 0 || d.nsegs < 2) return;
  const long total = (long)nlev * d.nstrips * d.nsegs;
  const long whole = total / round_waves, excess = total - whole * round_waves;
  if (whole < 1 || excess == 0 || excess * 4 > round_waves) return;   // the last round is at least a quarter full: leave it
  const int tjA = (rows + d.nsegs - 2) / (d.nsegs - 1);
  if ((rows + tjA - 1) / tjA != d.nsegs - 1) return;
  const long margin = round_waves / 64;   // a little room: the alt wavefronts run longer
  const int L = (int)((excess + margin + d.nstrips - 1) / d.nstrips);
  if (2 * L > nlev) return;
  d.alt_nk = L;
  d.alt_ng = d.nsegs - 1;
  d.alt_tj = tjA;
}

// ra_x = area + xfx(i) - xfx(i+1) is formed on the fly (sw_core.F90:908-917); so is ra_y.
struct MarchIn {
  vd qn, ar, cx, xf;  // row r:   q, area, crx, xfx
  vd cy, yf;          // face r-2: cry, yfx
};

// The register state of one fv_tp_2d march.  step(r) consumes row r; have_face says that face r-2 is
// wanted (r-2 >= jA), have_row that row r-3 is (r-3 >= jA); in the latter case fxv / fyv0 / fyv1 return
// 0.5*(fx + fx2)(j), 0.5*(fy + fy2)(j) and 0.5*(fy + fy2)(j+1) for j = r-3.
// UNI_AREA: the cell area is the same everywhere (Grid::geom == 2), so the area of row r-3 is the one of row r
template <int HORD, bool UNI_AREA = false>
struct Tp2dState {
  static constexpr int ORD_IN = (HORD == 10) ? 8 : HORD;  // tp_core.F90:136-141
  static constexpr int ORD_OU = HORD;
  PpmY<ORD_IN> ya;
  PpmY<ORD_OU> yb;
  vd fx2_0, fx2_1, fx2_2, fx2_3;  // fx2 of rows r, r-1, r-2, r-3
  vd fy2y_prev, yf_prev, fyv_prev;
  vd ar_1, ar_2, ar_3, cx_1, cx_2, cx_3;  // area and crx of rows r-1, r-2, r-3 (read once, used again at r-3)

  FV3_D void init() {
    ya.init();
    yb.init();
    fx2_0 = fx2_1 = fx2_2 = fx2_3 = vd(0.);
    fy2y_prev = yf_prev = fyv_prev = vd(0.);
    ar_1 = ar_2 = ar_3 = cx_1 = cx_2 = cx_3 = vd(1.);
  }
  FV3_D void step(const MarchIn &in, bool have_face, bool have_row, vd &fxv, vd &fyv0, vd &fyv1) {
    // ---- row r: inner x sweep and q_j --------------------------------------------------------------
    fx2_3 = fx2_2; fx2_2 = fx2_1; fx2_1 = fx2_0;
    const vd arj = UNI_AREA ? in.ar : ar_3, cxj = cx_3;  // rows r-3 (valid once three rows have been consumed)
    if constexpr (!UNI_AREA) { ar_3 = ar_2; ar_2 = ar_1; ar_1 = in.ar; }
    cx_3 = cx_2; cx_2 = cx_1; cx_1 = in.cx;
    fx2_0 = ppm_faces_x<ORD_IN>(in.qn, in.cx);
    const vd t = in.xf * fx2_0;
    const vd rax = in.ar + in.xf - shl1(in.xf);
    const vd qj = vdiv_r(in.qn * in.ar + t - shl1(t), rax, vrecip(rax));  // correctly rounded, 8 instructions instead of 11
    ya.push(in.qn);
    yb.push(qj);
    if (!have_face) return;
    // ---- face r-2: inner and outer y sweeps ------------------------------------------------------------
    const vd fy2 = ya.face(in.cy);
    const vd fyo = yb.face(in.cy);
    const vd fy2y = in.yf * fy2;
    const vd fyv = 0.5 * (fyo + fy2);
    // ---- row j = r-3: q_i and the outer x sweep ---------------------------------------------------------
    if (have_row) {
      const vd ray = arj + yf_prev - in.yf;
      const vd qi = vdiv_r(ya.row_m3() * arj + fy2y_prev - fy2y, ray, vrecip(ray));
      const vd fxo = ppm_faces_x<ORD_OU>(qi, cxj);
      fxv = 0.5 * (fxo + fx2_3);
      fyv0 = fyv_prev;
      fyv1 = fyv;
    }
    fy2y_prev = fy2y;
    yf_prev = in.yf;
    fyv_prev = fyv;
  }
};

// loads of one step except the transported field itself (crx/xfx: CX kind, cry/yfx: CY kind slabs)
FV3_D void march_load_metrics(MarchIn &in, const Grid &g, const StripGeom &s, int jA, int r, const double *crx,
                              const double *cry, const double *xfx, const double *yfx) {
  const int ilo = s.ilo;
  const long oCX = (long)g.iCX(ilo, r);
  in.ar = vload(g.area, (long)g.iA(ilo, r), s.A);
  in.cx = vload(crx, oCX, s.F);
  in.xf = vload(xfx, oCX, s.F);
  // rows before the segment's first face / cell are clamped: loaded but never used
  const int jf = (r - 2 < jA) ? jA : r - 2;
  const long oCY = (long)g.iCY(ilo, jf);
  in.cy = vload(cry, oCY, s.A);
  in.yf = vload(yfx, oCY, s.A);
}

// Row source of a stored field q (A kind slab).
struct FieldSrc {
  const Grid &g;
  const StripGeom &s;
  const double *q;
  using In = vd;
  FV3_D In load(int r) const { return vload(q, (long)g.iA(s.ilo, r), s.A); }
  FV3_D vd value(const In &in) const { return in; }
};

// a sink with `static constexpr bool kBranchFree = true` and row_bf(j, in, fx, fy0, fy1, on) gets the row step without control flow
template <class S, class = void>
struct sink_branch_free { static constexpr bool value = false; };
template <class S>
struct sink_branch_free<S, std::enable_if_t<S::kBranchFree>> { static constexpr bool value = true; };

// fv_tp_2d of the field produced row by row by `src`.  Software pipelining: the loads of step r+1 (and
// the sink's loads of its next row) are issued before the arithmetic of step r.
template <int HORD, class Src, class Sink>
FV3_D void tp2d_march_src(const Grid &g, const StripGeom &s, int jA, int jB, const Src &src, const double *crx,
                          const double *cry, const double *xfx, const double *yfx, Sink &sink) {
  Tp2dState<HORD> st;
  st.init();
  const int rlast = jB + 3;
  MarchIn nxt;
  march_load_metrics(nxt, g, s, jA, jA - 3, crx, cry, xfx, yfx);
  typename Src::In qnxt = src.load(jA - 3);
  typename Sink::In snxt = sink.load(jA);
  if constexpr (sink_branch_free<Sink>::value) {
    // no branch around a load or a store (a branch there costs an s_waitcnt vmcnt(0) at the top of every step, dsw_fused.h run_bf): the
    // warm-up steps load the sink's row jA again and hand it a dropped store (`on` = false)
    vd fxv(0.), fyv0(0.), fyv1(0.);
    vdrain_loads();
    for (int r = jA - 3; r <= rlast; r++) {
      MarchIn in = nxt;
      const typename Src::In qin = qnxt;
      const int rn = r < rlast ? r + 1 : rlast;
      march_load_metrics(nxt, g, s, jA, rn, crx, cry, xfx, yfx);
      qnxt = src.load(rn);
      in.qn = src.value(qin);
      const int j = r - 3;
      const bool on = j >= jA;
      st.step(in, r - 2 >= jA, on, fxv, fyv0, fyv1);
      const typename Sink::In sin = snxt;
      snxt = sink.load(on ? (j < jB ? j + 1 : jB) : jA);
      sink.row_bf(on ? j : jA, sin, fxv, fyv0, fyv1, on);
    }
    return;
  }
  for (int r = jA - 3; r <= rlast; r++) {
    MarchIn in = nxt;
    const typename Src::In qin = qnxt;
    const int rn = r < rlast ? r + 1 : rlast;
    march_load_metrics(nxt, g, s, jA, rn, crx, cry, xfx, yfx);
    qnxt = src.load(rn);
    in.qn = src.value(qin);
    const int j = r - 3;
    vd fxv, fyv0, fyv1;
    st.step(in, r - 2 >= jA, j >= jA, fxv, fyv0, fyv1);
    if (j >= jA) {
      const typename Sink::In sin = snxt;
      snxt = sink.load(j < jB ? j + 1 : jB);
      sink.row(j, sin, fxv, fyv0, fyv1);
    }
  }
}

template <int HORD, class Sink>
FV3_D void tp2d_march(const Grid &g, const StripGeom &s, int jA, int jB, const double *q, const double *crx,
                      const double *cry, const double *xfx, const double *yfx, Sink &sink) {
  const FieldSrc src{g, s, q};
  tp2d_march_src<HORD>(g, s, jA, jB, src, crx, cry, xfx, yfx, sink);
}

}  // namespace fv3
