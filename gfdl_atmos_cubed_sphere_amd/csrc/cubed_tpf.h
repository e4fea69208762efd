// cubed_tpf.h -- fv_tp_2d of up to three fields on the FRAME of a cubed-sphere face in ONE launch (model/tp_core.F90:85-241 with the
// face-edge forms; the hybrid path of fv3_api.hip: the marching kernels own the interior of a face, this kernel the frame).
//
// The pass kernels T1 / T2 / T3 of cubed_tp.h (one thread per point, every intermediate a round trip through HBM, one launch per
// pass and field: 9 launches for delp / w / pt) cost 1.0 ms of the 5.2 ms c_sw + d_sw pair on a C384 L127 face for 9 % of its points
// (VERDICT r2).  Here a workgroup owns a rectangle of the frame's flux points at one level, stages the field once in LDS (with the
// copy_corners maps of the x and of the y sweep applied while loading: two tiles that differ only in the corner regions of the halo),
// and runs inner sweeps -> q_i / q_j -> outer sweeps with the intermediates in LDS, field after field; the mass fluxes of the first
// field (delp) stay in LDS as the weights of the following ones (tp_core.F90:187-224).  Every face value is ppm_face_cs of cubed_tp.h on
// the same inputs, every q_i / q_j the expression of Tp2dCubedT2: the fluxes equal the passes' bit for bit.
#pragma once

#include "cubed_tp.h"

namespace fv3 {

constexpr int kTfT = 32;          // points of a tile along the face edge
constexpr int kTfMaxN = 42 * 14;  // doubles per LDS tile: (kTfT + 1 + 6 + pad) x (frame width + 1 + 6)
constexpr int kTfArrays = 8;      // qx, qy, fx2, fy2, q_i, q_j, mfx, mfy

struct TpfField {
  const double *q;   // A x nk
  double *fx, *fy;   // FX / FY outputs
  int hord;
};

struct Tp2dFrameFused {
  Grid g;
  TpfField f[3];
  int nf;                                // fields; f[0] is weighted with xfx / yfx, the others with f[0]'s fluxes (mass fluxes)
  const double *crx, *cry, *xfx, *yfx;   // CX / CY x nk
  const double *emfx, *emfy;             // FX / FY x nk or null: mass fluxes given by the caller weight EVERY field (tracer_2d)
  // deln_flux of the FIRST field (tp_core.F90:228-232: delp with nord_v / damp_v inside fv_tp_2d): its raw fluxes fx2 / fy2 (V / U
  // layout, DelnCubedL24) are added to f[0]'s fluxes on the levels with dcoef(k) > 1e-4, before they weight the other fields -- what
  // DelnCubedL5 does between two transports of the pass path.  null: no damping.
  const double *dfx = nullptr, *dfy = nullptr, *dcoef = nullptr;
  int w3;                                // width of the frame of flux points: i <= w3, i >= npx - w3, j <= w3, j >= npy - w3
  const int *klist;                      // levels of the launch (or null)
  int nS, nW;                            // tiles per south / north band, per west / east band
  int full;                              // 1: the whole face instead of the frame: nS tiles along i x nW bands of w3 rows

  FV3_HD int ntiles() const { return full ? nS * nW : 2 * nS + 2 * nW; }

  FV3_D void operator()(int b, int, int bz, int tid, double *lds) const {
    const int k = klist ? klist[bz] : bz;
    const int npx = g.npx, npy = g.npy;
    // the rectangle of flux points (i, j) in (is : ie+1, js : je+1) this workgroup owns
    int ia, ib, ja, jb;
    if (full) {
      const int tx = b % nS, ty = b / nS;
      ia = g.is + tx * kTfT;
      ib = ia + kTfT - 1 < g.ie + 1 ? ia + kTfT - 1 : g.ie + 1;
      ja = g.js + ty * w3;
      jb = ja + w3 - 1 < g.je + 1 ? ja + w3 - 1 : g.je + 1;
    } else if (b < 2 * nS) {
      const int t = b % nS;
      ia = g.is + t * kTfT;
      ib = ia + kTfT - 1 < g.ie + 1 ? ia + kTfT - 1 : g.ie + 1;
      if (b < nS) { ja = g.js; jb = w3; } else { ja = npy - w3; jb = g.je + 1; }
    } else {
      const int t = (b - 2 * nS) % nW;
      ja = w3 + 1 + t * kTfT;
      jb = ja + kTfT - 1 < npy - w3 - 1 ? ja + kTfT - 1 : npy - w3 - 1;
      if (b - 2 * nS < nW) { ia = g.is; ib = w3; } else { ia = npx - w3; ib = g.ie + 1; }
    }
    // tile of cells [i0, i1] x [j0, j1] = the rectangle + 3 (clipped to the arrays): everything any stage touches
    const int i0 = ia - 3 > g.isd ? ia - 3 : g.isd, i1 = ib + 3 < g.ied ? ib + 3 : g.ied;
    const int j0 = ja - 3 > g.jsd ? ja - 3 : g.jsd, j1 = jb + 3 < g.jed ? jb + 3 : g.jed;
    const int pw = i1 - i0 + 2, ph = j1 - j0 + 2;   // +1: face index ie+1 / je+1 of the flux tiles
    double *qx = lds, *qy = lds + kTfMaxN, *fx2 = lds + 2 * kTfMaxN, *fy2 = lds + 3 * kTfMaxN, *qi = lds + 4 * kTfMaxN,
           *qj = lds + 5 * kTfMaxN, *mfx = lds + 6 * kTfMaxN, *mfy = lds + 7 * kTfMaxN;
#define TF(a, i, j) (a)[((j) - j0) * pw + ((i) - i0)]
    const CA area = cview_A(g, g.area), dxa = cview_A(g, g.dxa), dya = cview_A(g, g.dya);
    const CA xf = cview_CX(g, xfx), yf = cview_CY(g, yfx), cx = cview_CX(g, crx), cy = cview_CY(g, cry);
    // (the barriers order LDS only: the fluxes a field stores are not read again here, so they drain behind the staging of the next one)
    for (int n = 0; n < nf; n++) {
      const CA q = cview_A(g, f[n].q);
      const int hord = f[n].hord, ord_in = (hord == 10) ? 8 : hord;
      // ---- stage the field: copy_corners (tp_core.F90:245-322) of the x sweep (dir 1) and of the y sweep (dir 2) as index maps ----
      for (int idx = tid; idx < (i1 - i0 + 1) * (j1 - j0 + 1); idx += kNT) {
        const int i = i0 + idx % (i1 - i0 + 1), j = j0 + idx / (i1 - i0 + 1);
        int ii = i, jj = j;
        copyc_src(1, npx, npy, ii, jj);
        TF(qx, i, j) = q(ii, jj, k);
        ii = i; jj = j;
        copyc_src(2, npx, npy, ii, jj);
        TF(qy, i, j) = q(ii, jj, k);
      }
      FV3_SYNC_LDS();
      // ---- T1: inner sweeps (tp_core.F90:143-168) on what T2 will read ----
      {
        // fx2(i, j): i in [ia, ib + 1] (faces of the q_j cells), j in [ja - 3, jb + 2] (the y lines of T3), inside (is:ie+1, jsd:jed)
        const int xa = ia, xb = (ib + 1 < g.ie + 1) ? ib + 1 : g.ie + 1, ya = (ja - 3 > g.jsd) ? ja - 3 : g.jsd, yb = (jb + 2 < g.jed) ? jb + 2 : g.jed;
        const int nx_ = xb - xa + 1, ny_ = yb - ya + 1;
        for (int idx = tid; idx < nx_ * ny_; idx += kNT) {
          const int i = xa + idx % nx_, j = ya + idx / nx_;
          auto ql = [&](int m) { return TF(qx, m, j); };
          auto dl = [&](int m) { return FV3_M(dxa, m, j); };
          TF(fx2, i, j) = ppm_face_cs(ql, dl, i, cx(i, j, k), ord_in, npx);
        }
        // fy2(i, j): i in [ia - 3, ib + 2], j in [ja, jb + 1], inside (isd:ied, js:je+1)
        const int ua = (ia - 3 > g.isd) ? ia - 3 : g.isd, ub = (ib + 2 < g.ied) ? ib + 2 : g.ied, va = ja, vb = (jb + 1 < g.je + 1) ? jb + 1 : g.je + 1;
        const int nu = ub - ua + 1, nv = vb - va + 1;
        for (int idx = tid; idx < nu * nv; idx += kNT) {
          const int i = ua + idx % nu, j = va + idx / nu;
          auto ql = [&](int m) { return TF(qy, i, m); };
          auto dl = [&](int m) { return FV3_M(dya, i, m); };
          TF(fy2, i, j) = ppm_face_cs(ql, dl, j, cy(i, j, k), ord_in, npy);
        }
      }
      FV3_SYNC_LDS();
      // ---- T2: q_i, q_j (:150-159, :171-178) ----
      {
        const int ua = (ia - 3 > g.isd) ? ia - 3 : g.isd, ub = (ib + 2 < g.ied) ? ib + 2 : g.ied, va = ja, vb = (jb < g.je) ? jb : g.je;
        const int nu = ub - ua + 1, nv = vb - va + 1;
        for (int idx = tid; idx < nu * nv; idx += kNT) {
          const int i = ua + idx % nu, j = va + idx / nu;
          const double y0 = yf(i, j, k), y1 = yf(i, j + 1, k), ar = FV3_M(area, i, j);
          const double fyy0 = y0 * TF(fy2, i, j), fyy1 = y1 * TF(fy2, i, j + 1);
          TF(qi, i, j) = (TF(qx, i, j) * ar + fyy0 - fyy1) / (ar + y0 - y1);
        }
        const int xa = ia, xb = (ib < g.ie) ? ib : g.ie, ya = (ja - 3 > g.jsd) ? ja - 3 : g.jsd, yb = (jb + 2 < g.jed) ? jb + 2 : g.jed;
        const int nx_ = xb - xa + 1, ny_ = yb - ya + 1;
        for (int idx = tid; idx < nx_ * ny_; idx += kNT) {
          const int i = xa + idx % nx_, j = ya + idx / nx_;
          const double x0 = xf(i, j, k), x1 = xf(i + 1, j, k), ar = FV3_M(area, i, j);
          const double fx10 = x0 * TF(fx2, i, j), fx11 = x1 * TF(fx2, i + 1, j);
          TF(qj, i, j) = (TF(qx, i, j) * ar + fx10 - fx11) / (ar + x0 - x1);
        }
      }
      FV3_SYNC_LDS();
      // ---- T3: outer sweeps, flux averaging and weighting (:161, :180, :187-224) on the rectangle ----
      {
        const int nx_ = ib - ia + 1, ny_ = jb - ja + 1;
        for (int idx = tid; idx < nx_ * ny_; idx += kNT) {
          const int i = ia + idx % nx_, j = ja + idx / nx_;
          if (j <= g.je) {
            auto ql = [&](int m) { return TF(qi, m, j); };
            auto dl = [&](int m) { return FV3_M(dxa, m, j); };
            const double fo = ppm_face_cs(ql, dl, i, cx(i, j, k), hord, npx);
            const double m = emfx ? cview_FX(g, emfx)(i, j, k) : ((n == 0) ? xf(i, j, k) : TF(mfx, i, j));
            double v = 0.5 * (fo + TF(fx2, i, j)) * m;
            if (n == 0 && dfx && dcoef[k] > 1.E-4) v = v + cview_V(g, dfx)(i, j, k);
            view_FX(g, f[n].fx)(i, j, k) = v;
            if (n == 0 && nf > 1) TF(mfx, i, j) = v;
          }
          if (i <= g.ie) {
            auto ql = [&](int m) { return TF(qj, i, m); };
            auto dl = [&](int m) { return FV3_M(dya, i, m); };
            const double fo = ppm_face_cs(ql, dl, j, cy(i, j, k), hord, npy);
            const double m = emfy ? cview_FY(g, emfy)(i, j, k) : ((n == 0) ? yf(i, j, k) : TF(mfy, i, j));
            double v = 0.5 * (fo + TF(fy2, i, j)) * m;
            if (n == 0 && dfy && dcoef[k] > 1.E-4) v = v + cview_U(g, dfy)(i, j, k);
            view_FY(g, f[n].fy)(i, j, k) = v;
            if (n == 0 && nf > 1) TF(mfy, i, j) = v;
          }
        }
      }
      FV3_SYNC_LDS();
    }
#undef TF
  }
};

}  // namespace fv3
