// cubed_common.h -- building blocks of the cubed-sphere (grid_type < 3, one tile per face) kernels.
//
// The face-edge and corner rules of the reference are index rules (i == 1, j == npy, ...), a few cells deep.  The kernels in
// cubed_csw.h / cubed_tp.h / cubed_dsw.h are written as short data-parallel PASSES over index boxes, one thread per
// (i, j, k) point, with the intermediates of a routine in context-owned device arrays (for the band of cells next to the
// face edges that these kernels are for, those arrays stay in L2 / Infinity Cache).  Every pass evaluates the reference's
// expressions in the reference's order, so the results are those of the row-wise formulation bit for bit.
#pragma once

#include "fv3_common.h"

namespace fv3 {

// extra members of the device gridstruct that only the cubed sphere needs (fv3_grid_upload_cubed)
struct CubedGeom {
  const double *edge_w, *edge_e, *edge_s, *edge_n;  // A -> B interpolation weights on the face edges, (npy) / (npx), 1-based
  const double *rsina;                              // (is:ie+1, js:je+1)
  const double *a11, *a12, *a21, *a22;              // cubed_to_latlon matrix, A layout; null = not uploaded
  const double *ec1, *ec2, *en1, *en2;              // adv_pe's unit vectors (3 planes each: A / FY / FX layouts); null = not uploaded
  double corner_f[12];                              // extrap_corner factors of a2b_ord4: sw, se, ne, nw x 3 pairs
  int ready;
};

// A pass = functor f(i, j, k) over the box [i0, i1] x [j0, j1] x [0, nk): 64 x `rows` points per workgroup, i fastest; a thread takes
// rows / 4 points (rows = 16: a quarter of the workgroups, the loads of a thread's four points in flight together, one halo row in 17
// instead of one in 5)
template <class F>
struct BoxPass {
  int i0, i1, j0, j1;
  F f;
  int rows = 4;
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *) const {
    for (int t = tid; t < 64 * rows; t += kNT) {
      const int i = i0 + bx * 64 + (t & 63), j = j0 + by * rows + (t >> 6);
      if (i <= i1 && j <= j1) f(i, j, bz);
    }
  }
};

// The same functor over the FRAME of a face only: the points of the box whose distance to a face edge is below w (i <= w,
// i >= npx - w, j <= w or j >= npy - w; the halo is part of the frame), for the level list klist[0 : nk) (null: identity).
// The south / north bands are full rows of the box; the west / east bands are the columns (i0 : w) and (npx - w : i1) of the
// rows between them, laid side by side in one 64-lane row.  Used by the hybrid path: the marching kernels own the interior
// of a face, these passes own the frame.
template <class F>
struct FramePass {
  int i0, i1, j0, j1, w, npx, npy;
  int nbx, nby_sn;   // workgroup columns / rows of the south + north bands; the west + east workgroups follow in blockIdx.x
  const int *klist;
  F f;
  int wc = 16;       // columns of a west / east workgroup's rows: 16 (x 16 rows), or 8 (x 32 rows) when neither band is wider than 8 points --
                     // a 7-point band then leaves 1 lane in 8 idle instead of 9 in 16
  FV3_HD void operator()(int bflat, int, int bz, int tid, double *) const {
    const int k = klist ? klist[bz] : bz;
    const int js1 = w < j1 ? w : j1, jn0 = (npy - w) > j0 ? (npy - w) : j0;   // south band j0..js1, north band jn0..j1
    const int ns = js1 - j0 + 1;
    const int nsn = nbx * nby_sn;
    // ONE call site of the functor for both kinds of workgroup (inlined twice -- once per branch, the second inside a loop -- the
    // heavier passes took 3 - 4 times the registers of their whole-face form, BoxPass: 158 against 54 for CswCubedP3, 195 against 55
    // for DswCubedD5): the point (i, j) of round r is formed first, then the functor runs.
    // South / north workgroup: 64 columns x 4 rows, one round.  West / east workgroup: 16 rows x 16 columns (32 x 8: wc) per round, the west
    // columns (i0 : w) or the east columns (npx - w : i1) of the rows between the bands.
    const bool sn = bflat < nsn;
    const int iw1 = w < i1 ? w : i1, ie0 = (npx - w) > i0 ? (npx - w) : i0;
    const int bw = bflat - nsn, side = bw & 1;
    const int ncol = sn ? 0 : (side ? i1 - ie0 + 1 : iw1 - i0 + 1);
    const int lg = wc == 8 ? 3 : 4;
    const int rounds = sn ? 1 : (ncol + wc - 1) / wc;
    for (int t = tid; t < 256; t += kNT) {
      for (int r = 0; r < rounds; r++) {
        int i, j;
        bool ok;
        if (sn) {
          const int bx = bflat % nbx, by = bflat / nbx;
          const int v = by * 4 + (t >> 6);
          j = v < ns ? j0 + v : jn0 + (v - ns);
          i = i0 + bx * 64 + (t & 63);
          ok = i <= i1 && j <= j1;
        } else {
          const int cc = (t & (wc - 1)) + wc * r;
          j = js1 + 1 + (bw >> 1) * (256 >> lg) + (t >> lg);
          i = (side ? ie0 : i0) + cc;
          ok = j < jn0 && cc < ncol;
        }
        if (ok) f(i, j, k);
      }
    }
  }
};
// a pass on the four corner squares of its box only: i within w of the west / east edge AND j within w of the south / north edge
// (one workgroup per corner and level; w <= 12).  Used where only copy_corners separates a pass chain from its LDS-tile form.
template <class F>
struct CornerPass {
  int i0, i1, j0, j1, w, npx, npy;
  const int *klist;
  F f;
  FV3_HD void operator()(int corner, int, int bz, int tid, double *) const {
    const int k = klist ? klist[bz] : bz;
    const bool east = corner & 1, north = corner & 2;
    const int ia = east ? ((npx - w) > i0 ? (npx - w) : i0) : i0, ib = east ? i1 : (w < i1 ? w : i1);
    const int ja = north ? ((npy - w) > j0 ? (npy - w) : j0) : j0, jb = north ? j1 : (w < j1 ? w : j1);
    const int ni = ib - ia + 1, nj = jb - ja + 1;
    for (int t = tid; t < ni * nj; t += kNT) f(ia + t % ni, ja + t / ni, k);
  }
};
// level list variant of BoxPass (the levels the hybrid path leaves to the full-face passes)
template <class F>
struct BoxPassK {
  int i0, i1, j0, j1;
  const int *klist;
  F f;
  FV3_HD void operator()(int bx, int by, int bz, int tid, double *) const {
    const int k = klist[bz];
    for (int t = tid; t < 256; t += kNT) {
      const int i = i0 + bx * 64 + (t & 63), j = j0 + by * 4 + (t >> 6);
      if (i <= i1 && j <= j1) f(i, j, k);
    }
  }
};

// Level views of the reference's array kinds (Fortran indices).
struct VA { double *p; int nid, isd, jsd; size_t n; FV3_HD double &operator()(int i, int j, int k) const { return p[(size_t)k * n + (size_t)(j - jsd) * nid + (i - isd)]; } };
struct CA { const double *p; int nid, isd, jsd; size_t n; FV3_HD double operator()(int i, int j, int k) const { return p[(size_t)k * n + (size_t)(j - jsd) * nid + (i - isd)]; } };
FV3_HD VA view_A(const Grid &g, double *p) { return VA{p, g.nid, g.isd, g.jsd, g.nA()}; }
FV3_HD VA view_U(const Grid &g, double *p) { return VA{p, g.nid, g.isd, g.jsd, g.nU()}; }
FV3_HD VA view_V(const Grid &g, double *p) { return VA{p, g.nid + 1, g.isd, g.jsd, g.nV()}; }
FV3_HD VA view_B(const Grid &g, double *p) { return VA{p, g.nid + 1, g.isd, g.jsd, g.nB()}; }
FV3_HD CA cview_A(const Grid &g, const double *p) { return CA{p, g.nid, g.isd, g.jsd, g.nA()}; }
FV3_HD CA cview_U(const Grid &g, const double *p) { return CA{p, g.nid, g.isd, g.jsd, g.nU()}; }
FV3_HD CA cview_V(const Grid &g, const double *p) { return CA{p, g.nid + 1, g.isd, g.jsd, g.nV()}; }
FV3_HD CA cview_B(const Grid &g, const double *p) { return CA{p, g.nid + 1, g.isd, g.jsd, g.nB()}; }
FV3_HD VA view_CX(const Grid &g, double *p) { return VA{p, g.nx + 1, g.is, g.jsd, g.nCX()}; }
FV3_HD VA view_CY(const Grid &g, double *p) { return VA{p, g.nid, g.isd, g.js, g.nCY()}; }
FV3_HD CA cview_CX(const Grid &g, const double *p) { return CA{p, g.nx + 1, g.is, g.jsd, g.nCX()}; }
FV3_HD CA cview_CY(const Grid &g, const double *p) { return CA{p, g.nid, g.isd, g.js, g.nCY()}; }
FV3_HD VA view_FX(const Grid &g, double *p) { return VA{p, g.nx + 1, g.is, g.js, g.nFX()}; }
FV3_HD VA view_FY(const Grid &g, double *p) { return VA{p, g.nx, g.is, g.js, g.nFY()}; }
FV3_HD CA cview_FX(const Grid &g, const double *p) { return CA{p, g.nx + 1, g.is, g.js, g.nFX()}; }
FV3_HD CA cview_FY(const Grid &g, const double *p) { return CA{p, g.nx, g.is, g.js, g.nFY()}; }
FV3_HD VA view_CC(const Grid &g, double *p) { return VA{p, g.nx, g.is, g.js, g.nCC()}; }
FV3_HD CA cview_CC(const Grid &g, const double *p) { return CA{p, g.nx, g.is, g.js, g.nCC()}; }
// 2-D metric arrays: k = 0
#define FV3_M(view, i, j) (view)((i), (j), 0)

// fill_4corners (sw_core.F90:3506-3553) as an index map: where the x- (dir 1) or y- (dir 2) sweep of c_sw / update_dz_c reads
// a cell of a corner region, it reads the cell this returns instead.  Every face owns all four corners.
FV3_HD void fill4_src(int dir, int npx, int npy, int &i, int &j) {
  if (dir == 1) {
    if (j == 0) {
      if (i == -1) { i = 0; j = 2; } else if (i == 0) { i = 0; j = 1; }                            // sw
      else if (i == npx + 1) { i = npx; j = 2; } else if (i == npx) { i = npx; j = 1; }           // se
    } else if (j == npy) {
      if (i == 0) { i = 0; j = npy - 1; } else if (i == -1) { i = 0; j = npy - 2; }                // nw
      else if (i == npx) { i = npx; j = npy - 1; } else if (i == npx + 1) { i = npx; j = npy - 2; }  // ne
    }
  } else {
    if (i == 0) {
      if (j == 0) { i = 1; j = 0; } else if (j == -1) { i = 2; j = 0; }                            // sw
      else if (j == npy) { i = 1; j = npy; } else if (j == npy + 1) { i = 2; j = npy; }           // nw
    } else if (i == npx) {
      if (j == 0) { i = npx - 1; j = 0; } else if (j == -1) { i = npx - 2; j = 0; }                // se
      else if (j == npy) { i = npx - 1; j = npy; } else if (j == npy + 1) { i = npx - 2; j = npy; }  // ne
    }
  }
}

// copy_corners (tp_core.F90:245-322) as an index map: the cell a sweep in direction dir reads in place of a cell of a
// corner region of the halo (ng = 3).
FV3_HD void copyc_src(int dir, int npx, int npy, int &i, int &j) {
  const bool w = i <= 0, e = i >= npx, s = j <= 0, n = j >= npy;
  if (!((w || e) && (s || n))) return;
  const int ii = i, jj = j;
  if (dir == 1) {
    if (w && s) { i = jj; j = 1 - ii; }
    else if (e && s) { i = npy - jj; j = ii - npx + 1; }
    else if (e && n) { i = jj; j = 2 * npx - 1 - ii; }
    else { i = npy - jj; j = ii - 1 + npx; }
  } else {
    if (w && s) { i = 1 - jj; j = ii; }
    else if (e && s) { i = npy + jj - 1; j = npx - ii; }
    else if (e && n) { i = 2 * npy - 1 - jj; j = ii; }
    else { i = jj + 1 - npx; j = npy - ii; }
  }
}

// edge_interpolate4, sw_core.F90:3348-3359
FV3_HD double edge_interpolate4(double u1, double u2, double u3, double u4, double d1, double d2, double d3, double d4) {
  const double t1 = d1 + d2, t2 = d3 + d4;
  return 0.5 * (((t1 + d2) * u2 - d2 * u1) / t1 + ((t2 + d3) * u3 - d3 * u4) / t2);
}

}  // namespace fv3
