// tracer_kernels.h -- tracer_2d (model/fv_tracer2d.F90:297-557): sub-cycled transport of nq tracers with the
// mass fluxes (mfx, mfy) and Courant numbers (cx, cy) accumulated over the acoustic substeps.
//   TracerPrep  : area fluxes xfx, yfx and the per-level maximum Courant number cmax(k)   (:362-400)
//   TracerScale : cx, xfx, mfx, cy, yfx, mfy *= frac(k)                                   (:421-456)
//   TracerStep  : one sub-cycle: dp2, fv_tp_2d of every tracer with mass-flux weighting, flux-form update
//                 (:481-531).  q -> q_out (ping-pong: neighbour tiles still read the old halo).
// The global maximum over ranks (mp_reduce_max, :405) and the choice of nsplt / ksplt / frac stay on the host.
#pragma once

#include "fv3_common.h"
#include "tp2d_tile.h"

namespace fv3 {

struct TracerPrep {
  Grid g;
  int npz, q_split;
  const double *cx, *cy;
  double *xfx, *yfx;
  double *cmax;  // device, npz, zero-initialised; max taken over non-negative values
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *lds) const {
    const int k = bz;
    const size_t oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const int ncx = (int)g.nCX(), ncy = (int)g.nCY();
    double lmax = 0.;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH; idx += kNT) {
      if (idx < ncx) {
        const int i = g.is + idx % (g.nx + 1), j = g.jsd + idx / (g.nx + 1);
        const double c = cx[oCX + idx];
        xfx[oCX + idx] = (c > 0.) ? c * g.dxa[g.iA(i - 1, j)] * g.dy[g.iV(i, j)] * g.sinsg(i - 1, j, 3)
                                  : c * g.dxa[g.iA(i, j)] * g.dy[g.iV(i, j)] * g.sinsg(i, j, 1);
      }
      if (idx < ncy) {
        const int i = g.isd + idx % g.nid, j = g.js + idx / g.nid;
        const double c = cy[oCY + idx];
        yfx[oCY + idx] = (c > 0.) ? c * g.dya[g.iA(i, j - 1)] * g.dx[g.iU(i, j)] * g.sinsg(i, j - 1, 4)
                                  : c * g.dya[g.iA(i, j)] * g.dx[g.iU(i, j)] * g.sinsg(i, j, 2);
      }
      if (q_split == 0 && idx < g.nx * g.ny) {
        const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
        const double a = fabs(cx[oCX + g.iCX(i, j)]), b = fabs(cy[oCY + g.iCY(i, j)]);
        double v = dmax(a, b);
        if (!(k + 1 < npz / 6)) v = v + 1. - g.sinsg(i, j, 5);
        lmax = dmax(lmax, v);
      }
    }
    if (q_split == 0) {
#ifdef FV3_HOST_EMU
      (void)lds;
      if (lmax > cmax[k]) cmax[k] = lmax;
#else
      // non-negative doubles order like their bit patterns
      atomicMax((unsigned long long *)&cmax[k], (unsigned long long)__double_as_longlong(lmax));
#endif
    }
  }
};

struct TracerScale {
  Grid g;
  const double *frac;  // device, npz
  double *cx, *xfx, *mfx, *cy, *yfx, *mfy;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const int k = bz;
    const double f = frac[k];
    const int ncx = (int)g.nCX(), ncy = (int)g.nCY(), nfx = (int)g.nFX(), nfy = (int)g.nFY();
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH; idx += kNT) {
      if (idx < ncx) {
        cx[(size_t)k * ncx + idx] = cx[(size_t)k * ncx + idx] * f;
        xfx[(size_t)k * ncx + idx] = xfx[(size_t)k * ncx + idx] * f;
      }
      if (idx < ncy) {
        cy[(size_t)k * ncy + idx] = cy[(size_t)k * ncy + idx] * f;
        yfx[(size_t)k * ncy + idx] = yfx[(size_t)k * ncy + idx] * f;
      }
      if (idx < nfx) mfx[(size_t)k * nfx + idx] = mfx[(size_t)k * nfx + idx] * f;
      if (idx < nfy) mfy[(size_t)k * nfy + idx] = mfy[(size_t)k * nfy + idx] * f;
    }
  }
};

template <int TI, int TJ>
struct TracerStep {
  Grid g;
  int npz, nq, it, nsplt, hord, nord_tr;
  double trdm;
  const int *ksplt;  // device, npz
  const double *q, *dp1, *mfx, *mfy, *cx, *cy, *xfx, *yfx;
  double *q_out, *dp1_out;
  const double *mass = nullptr;   // deln_flux's mass (null: dp1, fv_tracer2d.F90:497-505; d_sw's inline_q passes its delp, sw_core.F90:1034)
  using TS = Tp2dScratch<TI, TJ>;
  using DS = DelnScratch<TI, TJ>;
  static constexpr int nQ = (TI + 6) * (TJ + 6);
  static constexpr int nScr = TS::total > DS::total ? TS::total : DS::total;
  static constexpr int nFXt = (TI + 1) * TJ, nFYt = TI * (TJ + 1);
  static constexpr int lds_doubles = 2 * nQ + nScr + nFXt + nFYt;
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *lds) const {
    const int k = bz;
    const TileBox b = make_box<TI, TJ>(g, bx, by);
    const int i0 = b.i0, j0 = b.j0;
    const size_t nA = g.nA();
    const size_t oA = (size_t)k * nA, oCX = (size_t)k * g.nCX(), oCY = (size_t)k * g.nCY();
    const size_t oFX = (size_t)k * g.nFX(), oFY = (size_t)k * g.nFY();
    const bool active = it <= ksplt[k];
    if (!active) {  // the level is finished: carry q (and dp1) over to the output buffers
      FV3_TILE_FOR(TI, TJ, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        for (int iq = 0; iq < nq; iq++)
          q_out[((size_t)iq * npz + k) * nA + g.iA(i, j)] = q[((size_t)iq * npz + k) * nA + g.iA(i, j)];
        if (it != nsplt) dp1_out[oA + g.iA(i, j)] = dp1[oA + g.iA(i, j)];
      }
      return;
    }
    double *p = lds;
    const Tile sq{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;
    const Tile sm{p, i0 - 3, j0 - 3, TI + 6}; p += nQ;
    double *scr = p; p += nScr;
    const Tile sfx{p, i0, j0, TI + 1}; p += nFXt;
    const Tile sfy{p, i0, j0, TI}; p += nFYt;
    const bool damp = (it == 1 && trdm > 1.e-4);
    if (damp) load_tile<TI + 6, TJ + 6>(sm, (mass ? mass : dp1) + oA, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
    for (int iq = 0; iq < nq; iq++) {
      const double *qk = q + ((size_t)iq * npz + k) * nA;
      FV3_SYNC();
      load_tile<TI + 6, TJ + 6>(sq, qk, g.nid, g.isd, g.ied, g.jsd, g.jed, tid);
      FV3_SYNC();
      tp2d_tile<TI, TJ>(g, b, tid, sq, cx + oCX, cy + oCY, xfx + oCX, yfx + oCY, nullptr, nullptr, hord, scr, sfx, sfy);
      FV3_TILE_FOR((TI + 1), TJ, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast + 1 || j > b.jlast) continue;
        sfx(i, j) = sfx(i, j) * mfx[oFX + g.iFX(i, j)];
      }
      FV3_TILE_FOR(TI, (TJ + 1), li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast + 1) continue;
        sfy(i, j) = sfy(i, j) * mfy[oFY + g.iFY(i, j)];
      }
      FV3_SYNC();
      if (damp) {
        const double dmp = ipow(trdm * g.da_min, nord_tr + 1);
        Tile fxd, fyd;
        deln_tile<TI, TJ>(g, b, tid, sq, nord_tr, dmp, false, scr, fxd, fyd);
        const double damp2 = 0.5 * dmp;
        FV3_TILE_FOR((TI + 1), TJ, li_, lj_) {
          const int i = i0 + li_, j = j0 + lj_;
          if (i > b.ilast + 1 || j > b.jlast) continue;
          sfx(i, j) = sfx(i, j) + damp2 * (sm(i - 1, j) + sm(i, j)) * fxd(i, j);
        }
        FV3_TILE_FOR(TI, (TJ + 1), li_, lj_) {
          const int i = i0 + li_, j = j0 + lj_;
          if (i > b.ilast || j > b.jlast + 1) continue;
          sfy(i, j) = sfy(i, j) + damp2 * (sm(i, j - 1) + sm(i, j)) * fyd(i, j);
        }
        FV3_SYNC();
      }
      FV3_TILE_FOR(TI, TJ, li_, lj_) {
        const int i = i0 + li_, j = j0 + lj_;
        if (i > b.ilast || j > b.jlast) continue;
        const double ra = g.rarea[g.iA(i, j)], d1 = dp1[oA + g.iA(i, j)];
        const double dp2 = d1 + (mfx[oFX + g.iFX(i, j)] - mfx[oFX + g.iFX(i + 1, j)] + mfy[oFY + g.iFY(i, j)] -
                                 mfy[oFY + g.iFY(i, j + 1)]) * ra;
        q_out[((size_t)iq * npz + k) * nA + g.iA(i, j)] =
            (sq(i, j) * d1 + (sfx(i, j) - sfx(i + 1, j) + sfy(i, j) - sfy(i, j + 1)) * ra) / dp2;
        if (iq == nq - 1 && it != nsplt) dp1_out[oA + g.iA(i, j)] = dp2;
      }
    }
  }
};

// sw_core.F90:1020-1043 (inline_q): d_sw's fv_tp_2d of a tracer gets mass = delp AFTER the compute domain was updated (:1021-1024)
// and before any halo update: new values inside, the old ones in the halo
struct InlineQMass {
  Grid g;
  const double *delp_old, *delp_new;
  double *mass;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const size_t o = (size_t)bz * g.nA();
    const int n = g.nid * g.njd;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.isd + idx % g.nid, j = g.jsd + idx / g.nid;
      const bool in = i >= g.is && i <= g.ie && j >= g.js && j <= g.je;
      mass[o + idx] = in ? delp_new[o + idx] : delp_old[o + idx];
    }
  }
};

// mfx = mfx + fx, mfy = mfy + fy (sw_core.F90:949-962 for a d_sw that was handed zeroed flux arrays of its own)
struct FluxAccum {
  Grid g;
  double *mfx, *mfy;
  const double *fx, *fy;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const size_t nx_ = g.nFX(), ny_ = g.nFY();
    for (size_t idx = (size_t)bx * CH + tid; idx < (size_t)(bx + 1) * CH; idx += kNT) {
      if (idx < nx_) mfx[(size_t)bz * nx_ + idx] = mfx[(size_t)bz * nx_ + idx] + fx[(size_t)bz * nx_ + idx];
      if (idx < ny_) mfy[(size_t)bz * ny_ + idx] = mfy[(size_t)bz * ny_ + idx] + fy[(size_t)bz * ny_ + idx];
    }
  }
};

// fill2D (model/fv_fill.F90:183-258)
struct Fill2dMass {   // :228-235
  Grid g;
  const double *q, *delp;
  double *qt;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const size_t o = (size_t)bz * g.nA();
    const int n = g.nx * g.ny;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int a = g.iA(g.is + idx % g.nx, g.js + idx / g.nx);
      qt[o + a] = q[o + a] * delp[o + a] * g.area[a];
    }
  }
};
struct Fill2dApply {  // :238-256
  Grid g;
  const double *qt, *delp;
  double *q;
  static constexpr int CH = 1024;
  FV3_HD void operator()(int bx, int, int bz, int tid, double *) const {
    const size_t o = (size_t)bz * g.nA();
    const int n = g.nx * g.ny;
    for (int idx = bx * CH + tid; idx < (bx + 1) * CH && idx < n; idx += kNT) {
      const int i = g.is + idx % g.nx, j = g.js + idx / g.nx;
      const double c0 = qt[o + g.iA(i, j)], w = qt[o + g.iA(i - 1, j)], e = qt[o + g.iA(i + 1, j)], s_ = qt[o + g.iA(i, j - 1)],
                   n_ = qt[o + g.iA(i, j + 1)];
      const double fxw = (w * c0 < 0.) ? w - c0 : 0., fxe = (c0 * e < 0.) ? c0 - e : 0.;
      const double fys = (s_ * c0 < 0.) ? s_ - c0 : 0., fyn = (c0 * n_ < 0.) ? c0 - n_ : 0.;
      const int a = g.iA(i, j);
      q[o + a] = q[o + a] + 0.25 * (fxw - fxe + fys - fyn) / (delp[o + a] * g.area[a]);
    }
  }
};

// prt_maxmin / prt_mxm (tools/fv_diagnostics.F90:4213-4313): minimum and maximum of one level of an A-kind field over the compute
// domain, one workgroup per level; out[2 bz] = min, out[2 bz + 1] = max
struct LevelMinMax {
  Grid g;
  const double *q;
  double *out;
  FV3_HD void operator()(int, int, int bz, int tid, double *lds) const {
    const double *s = q + (size_t)bz * g.nA();
    const int n = g.nx * g.ny;
    double lo = s[g.iA(g.is, g.js)], hi = lo;
    for (int idx = tid; idx < n; idx += kNT) {
      const double v = s[g.iA(g.is + idx % g.nx, g.js + idx / g.nx)];
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    lds[tid] = lo;
    lds[kNT + tid] = hi;
    FV3_SYNC();
    for (int st = kNT / 2; st > 0; st >>= 1) {
      if (tid < st) {
        lds[tid] = lds[tid + st] < lds[tid] ? lds[tid + st] : lds[tid];
        lds[kNT + tid] = lds[kNT + tid + st] > lds[kNT + tid] ? lds[kNT + tid + st] : lds[kNT + tid];
      }
      FV3_SYNC();
    }
    if (tid == 0) {
      out[2 * bz] = lds[0];
      out[2 * bz + 1] = lds[kNT];
    }
  }
};

}  // namespace fv3
