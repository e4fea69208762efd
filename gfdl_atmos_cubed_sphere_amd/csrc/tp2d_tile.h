// tp2d_tile.h -- workgroup-level building blocks shared by the transport kernels:
//   * load_tile      : stage a halo'd patch of one k-slab into LDS
//   * tp2d_tile      : fv_tp_2d (model/tp_core.F90:85-241) for one TI x TJ tile: the Lin-Rood
//                      inner/outer PPM sweeps with q_i, q_j, fx2, fy2 living only in LDS
//   * deln_tile      : the iterated-Laplacian diffusive fluxes of deln_flux
//                      (tp_core.F90:1267-1447) == del6_vt_flux (sw_core.F90:1608-1737)
// Notation: E(n) = the tile's cells expanded by n cells on every side.
#pragma once

#include "fv3_common.h"
#include "ppm.h"

namespace fv3 {

// Limits of a tile clipped to the compute domain.
struct TileBox {
  int i0, j0;      // first cell
  int ilast, jlast;  // last cell inside the compute domain (<= ie, je)
};

template <int TI, int TJ>
FV3_HD TileBox make_box(const Grid &g, int bx, int by) {
  TileBox b;
  b.i0 = g.is + bx * TI;
  b.j0 = g.js + by * TJ;
  b.ilast = (b.i0 + TI - 1 < g.ie) ? b.i0 + TI - 1 : g.ie;
  b.jlast = (b.j0 + TJ - 1 < g.je) ? b.j0 + TJ - 1 : g.je;
  return b;
}

// Load src (one k-slab with origin (ilo,jlo) and leading dimension ld, defined on
// ilo..ihi x jlo..jhi) on the index box [ia,ib] x [ja,jb] into tile t; points outside the
// array are set to 0 (they are never used for a value that is kept).
template <int W, int H>
FV3_HD void load_tile(const Tile &t, const double *src, int ld, int ilo, int ihi, int jlo, int jhi, int tid) {
  FV3_TILE_FOR(W, H, li, lj) {
    const int i = t.i0 + li, j = t.j0 + lj;
    double v = 0.;
    if (i >= ilo && i <= ihi && j >= jlo && j <= jhi) v = src[(j - jlo) * ld + (i - ilo)];
    t.p[lj * t.pitch + li] = v;
  }
}

// LDS footprint (doubles) of tp2d_tile's private scratch (fy2, fx2, q_i, q_j).
template <int TI, int TJ>
struct Tp2dScratch {
  static constexpr int WQ = TI + 6, HQ = TJ + 6;
  static constexpr int nFY2 = WQ * (TJ + 1), nFX2 = (TI + 1) * HQ, nQI = WQ * TJ, nQJ = TI * HQ;
  static constexpr int total = nFY2 + nFX2 + nQI + nQJ;
};

// fv_tp_2d for one tile.  On entry sq holds q on E(3) (synchronised).  crx/xfx (CX kind) and
// cry/yfx (CY kind) are the global slabs of this level; ra_x/ra_y may be null, in which case
// they are formed as area + xfx(i) - xfx(i+1) (sw_core.F90:908-917).
// On exit (after the trailing barrier) sfx(i,j), i in [i0, ilast+1], j in [j0, jlast] holds
// 0.5*(fx + fx2) and sfy(i,j), i in [i0, ilast], j in [j0, jlast+1] holds 0.5*(fy + fy2):
// the face values of tp_core.F90:193/198/219/224 before the multiplication by the mass flux.
template <int TI, int TJ>
FV3_HD void tp2d_tile(const Grid &g, const TileBox &b, int tid, const Tile &sq, const double *crx,
                      const double *cry, const double *xfx, const double *yfx, const double *ra_x,
                      const double *ra_y, int hord, double *scratch, const Tile &sfx, const Tile &sfy) {
  using S = Tp2dScratch<TI, TJ>;
  constexpr int WQ = S::WQ, HQ = S::HQ;
  const int i0 = b.i0, j0 = b.j0;
  const Tile sfy2{scratch, i0 - 3, j0, WQ};                               // (E3 in i) x faces j0..j0+TJ
  const Tile sfx2{scratch + S::nFY2, i0, j0 - 3, TI + 1};                 // faces i0..i0+TI x (E3 in j)
  const Tile sqi{scratch + S::nFY2 + S::nFX2, i0 - 3, j0, WQ};            // (E3 in i) x tile rows
  const Tile sqj{scratch + S::nFY2 + S::nFX2 + S::nQI, i0, j0 - 3, TI};   // tile cols x (E3 in j)
  const int ord_in = (hord == 10) ? 8 : hord;  // tp_core.F90:136-141
  const int ord_ou = hord;
  const double lim = g.lim_fac;

  // S1: inner sweeps on the unmodified field (tp_core.F90:147 and :168)
  FV3_TILE_FOR(WQ, (WQ * (TJ + 1)) / WQ, li_, lj_) {
    const int i = i0 - 3 + li_, j = j0 + lj_;
    if (i > b.ilast + 3 || j > b.jlast + 1) continue;
    sfy2(i, j) = ppm_face_tp(&sq(i, j), sq.pitch, cry[g.iCY(i, j)], ord_in, lim);
  }
  FV3_TILE_FOR((TI + 1), ((TI + 1) * HQ) / (TI + 1), li_, lj_) {
    const int i = i0 + li_, j = j0 - 3 + lj_;
    if (i > b.ilast + 1 || j > b.jlast + 3) continue;
    sfx2(i, j) = ppm_face_tp(&sq(i, j), 1, crx[g.iCX(i, j)], ord_in, lim);
  }
  FV3_SYNC();
  // S2: intermediate fields q_i (:150-159) and q_j (:171-178)
  FV3_TILE_FOR(WQ, (WQ * TJ) / WQ, li_, lj_) {
    const int i = i0 - 3 + li_, j = j0 + lj_;
    if (i > b.ilast + 3 || j > b.jlast) continue;
    const double y0 = yfx[g.iCY(i, j)], y1 = yfx[g.iCY(i, j + 1)], ar = g.area[g.iA(i, j)];
    const double fyy0 = y0 * sfy2(i, j), fyy1 = y1 * sfy2(i, j + 1);
    const double ray = ra_y ? ra_y[g.iRY(i, j)] : (ar + y0 - y1);
    sqi(i, j) = (sq(i, j) * ar + fyy0 - fyy1) / ray;
  }
  FV3_TILE_FOR(TI, (TI * HQ) / TI, li_, lj_) {
    const int i = i0 + li_, j = j0 - 3 + lj_;
    if (i > b.ilast || j > b.jlast + 3) continue;
    const double x0 = xfx[g.iCX(i, j)], x1 = xfx[g.iCX(i + 1, j)], ar = g.area[g.iA(i, j)];
    const double fx10 = x0 * sfx2(i, j), fx11 = x1 * sfx2(i + 1, j);
    const double rax = ra_x ? ra_x[g.iRX(i, j)] : (ar + x0 - x1);
    sqj(i, j) = (sq(i, j) * ar + fx10 - fx11) / rax;
  }
  FV3_SYNC();
  // S3: outer sweeps (:161, :180) and flux averaging
  FV3_TILE_FOR((TI + 1), ((TI + 1) * TJ) / (TI + 1), li_, lj_) {
    const int i = i0 + li_, j = j0 + lj_;
    if (i > b.ilast + 1 || j > b.jlast) continue;
    const double f = ppm_face_tp(&sqi(i, j), 1, crx[g.iCX(i, j)], ord_ou, lim);
    sfx(i, j) = 0.5 * (f + sfx2(i, j));
  }
  FV3_TILE_FOR(TI, (TI * (TJ + 1)) / TI, li_, lj_) {
    const int i = i0 + li_, j = j0 + lj_;
    if (i > b.ilast || j > b.jlast + 1) continue;
    const double f = ppm_face_tp(&sqj(i, j), sqj.pitch, cry[g.iCY(i, j)], ord_ou, lim);
    sfy(i, j) = 0.5 * (f + sfy2(i, j));
  }
  FV3_SYNC();
}

// LDS footprint of deln_tile: d2 on E(3), x-fluxes on (TI+7)x(TJ+6), y-fluxes on (TI+6)x(TJ+7).
template <int TI, int TJ>
struct DelnScratch {
  static constexpr int nD2 = (TI + 6) * (TJ + 6), nFX = (TI + 7) * (TJ + 6), nFY = (TI + 6) * (TJ + 7);
  static constexpr int total = nD2 + nFX + nFY;
};

// Diffusive del-(2*nord+2) fluxes of a cell-centred field held in sq on E(3) (only E(1+nord)
// is read).  premul: d2 = damp*q (deln_flux without mass, del6_vt_flux) or d2 = q (deln_flux
// with mass).  On exit fxd(i,j), i in [i0, ilast+1], j in [j0, jlast] and fyd(i,j), i in
// [i0, ilast], j in [j0, jlast+1] hold fx2 / fy2 of tp_core.F90:1321-1382.  nord <= 2.
template <int TI, int TJ>
FV3_HD void deln_tile(const Grid &g, const TileBox &b, int tid, const Tile &sq, int nord, double damp,
                      bool premul, double *scratch, Tile &fxd, Tile &fyd) {
  using S = DelnScratch<TI, TJ>;
  const int i0 = b.i0, j0 = b.j0;
  const Tile d2{scratch, i0 - 3, j0 - 3, TI + 6};
  fxd = Tile{scratch + S::nD2, i0 - 3, j0 - 3, TI + 7};
  fyd = Tile{scratch + S::nD2 + S::nFX, i0 - 3, j0 - 3, TI + 6};
  const int il = b.ilast, jl = b.jlast;
  {
    const int e = 1 + nord;
    FV3_TILE_FOR((TI + 6), ((TI + 6) * (TJ + 6)) / (TI + 6), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
      if (i < i0 - e || i > il + e || j < j0 - e || j > jl + e) continue;
      d2(i, j) = premul ? damp * sq(i, j) : sq(i, j);
    }
  }
  FV3_SYNC();
  FV3_TILE_FOR((TI + 7), ((TI + 7) * (TJ + 6)) / (TI + 7), li_, lj_) {
    const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
    if (i < i0 - nord || i > il + nord + 1 || j < j0 - nord || j > jl + nord) continue;
    fxd(i, j) = g.del6_v[g.iV(i, j)] * (d2(i - 1, j) - d2(i, j));
  }
  FV3_TILE_FOR((TI + 6), ((TI + 6) * (TJ + 7)) / (TI + 6), li_, lj_) {
    const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
    if (i < i0 - nord || i > il + nord || j < j0 - nord || j > jl + nord + 1) continue;
    fyd(i, j) = g.del6_u[g.iU(i, j)] * (d2(i, j - 1) - d2(i, j));
  }
  FV3_SYNC();
  for (int n = 1; n <= nord; n++) {
    const int nt = nord - n;
    FV3_TILE_FOR((TI + 6), ((TI + 6) * (TJ + 6)) / (TI + 6), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
      if (i < i0 - nt - 1 || i > il + nt + 1 || j < j0 - nt - 1 || j > jl + nt + 1) continue;
      d2(i, j) = (fxd(i, j) - fxd(i + 1, j) + fyd(i, j) - fyd(i, j + 1)) * g.rarea[g.iA(i, j)];
    }
    FV3_SYNC();
    FV3_TILE_FOR((TI + 7), ((TI + 7) * (TJ + 6)) / (TI + 7), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
      if (i < i0 - nt || i > il + nt + 1 || j < j0 - nt || j > jl + nt) continue;
      fxd(i, j) = g.del6_v[g.iV(i, j)] * (d2(i, j) - d2(i - 1, j));
    }
    FV3_TILE_FOR((TI + 6), ((TI + 6) * (TJ + 7)) / (TI + 6), li_, lj_) {
      const int i = i0 - 3 + li_, j = j0 - 3 + lj_;
      if (i < i0 - nt || i > il + nt || j < j0 - nt || j > jl + nt + 1) continue;
      fyd(i, j) = g.del6_u[g.iU(i, j)] * (d2(i, j) - d2(i, j - 1));
    }
    FV3_SYNC();
  }
}

}  // namespace fv3
