// cubed_a2b.h -- a2b_ord4 (model/a2b_edge.F90:47-327) on a cubed-sphere face: 4th-order interpolation of a cell-centred field to
// the cell corners with the one-sided forms next to the face edges, the two-sided edge values weighted by edge_w/e/s/n and
// the three-way extrapolation at the cube corners (extrap_corner :452-462).  Two passes for up to four fields at a time (the
// interface of A2BCorners in nh_kernels.h: corner (i, j) of level k stored at iA(i, j) of an A-layout slab):
//   Pa  qx on (1:npx, 1:npy-1), qy on (1:npx-1, 1:npy), the corner values on the four edges and corners of the face
//   Pb  the interior corners (2:npx-1, 2:npy-1)
#pragma once

#include "cubed_common.h"

namespace fv3 {

struct A2bCubedState {
  Grid g;
  CubedGeom cg;
  const double *in[4];
  double *out[4];
  double *qx[4], *qy[4];  // scratch, A layout
  int nlev[4];
  int nf;
  double scale[4];
  double top[4];
  int override_mask;
};

struct A2bCubedPa {
  A2bCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    constexpr double b1 = 7. / 12., b2 = -1. / 12., r3 = 1. / 3.;
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    const CA dxa = cview_A(g, g.dxa), dya = cview_A(g, g.dya);
    for (int f = 0; f < s.nf; f++) {
      if (k >= s.nlev[f]) continue;
      if (k == 0 && ((s.override_mask >> f) & 1)) continue;
      const CA qin_ = cview_A(g, s.in[f]);
      const double sc = s.scale[f];
      auto q = [&](int ii, int jj) { return sc != 1.0 ? qin_(ii, jj, k) * sc : qin_(ii, jj, k); };
      // ---- qx(i, j), i = 1..npx, j = 1..npy-1 (:133-173)
      if (j <= npy - 1) {
        auto qx_edge = [&](int e, int in1, int in2, int ou1, int ou2) {  // i = 1: (1, 2 | 0, -1); i = npx: (npx-1, npx-2 | npx, npx+1)
          const double g_in = FV3_M(dxa, in2, j) / FV3_M(dxa, in1, j), g_ou = FV3_M(dxa, ou2, j) / FV3_M(dxa, ou1, j);
          (void)e;
          return 0.5 * (((2. + g_in) * q(in1, j) - q(in2, j)) / (1. + g_in) + ((2. + g_ou) * q(ou1, j) - q(ou2, j)) / (1. + g_ou));
        };
        auto qx_std = [&](int ii) { return b2 * (q(ii - 2, j) + q(ii + 1, j)) + b1 * (q(ii - 1, j) + q(ii, j)); };
        double v;
        if (i == 1)
          v = qx_edge(1, 1, 2, 0, -1);
        else if (i == npx)
          v = qx_edge(npx, npx - 1, npx - 2, npx, npx + 1);
        else if (i == 2) {
          const double g_in = FV3_M(dxa, 2, j) / FV3_M(dxa, 1, j);
          v = (3. * (g_in * q(1, j) + q(2, j)) - (g_in * qx_edge(1, 1, 2, 0, -1) + qx_std(3))) / (2. + 2. * g_in);
        } else if (i == npx - 1) {
          const double g_in = FV3_M(dxa, npx - 2, j) / FV3_M(dxa, npx - 1, j);
          v = (3. * (q(npx - 2, j) + g_in * q(npx - 1, j)) - (g_in * qx_edge(npx, npx - 1, npx - 2, npx, npx + 1) + qx_std(npx - 2))) / (2. + 2. * g_in);
        } else
          v = qx_std(i);
        view_A(g, s.qx[f])(i, j, k) = v;
      }
      // ---- qy(i, j), i = 1..npx-1, j = 1..npy (:183-222)
      if (i <= npx - 1) {
        auto qy_edge = [&](int in1, int in2, int ou1, int ou2) {
          const double g_in = FV3_M(dya, i, in2) / FV3_M(dya, i, in1), g_ou = FV3_M(dya, i, ou2) / FV3_M(dya, i, ou1);
          return 0.5 * (((2. + g_in) * q(i, in1) - q(i, in2)) / (1. + g_in) + ((2. + g_ou) * q(i, ou1) - q(i, ou2)) / (1. + g_ou));
        };
        auto qy_std = [&](int jj) { return b2 * (q(i, jj - 2) + q(i, jj + 1)) + b1 * (q(i, jj - 1) + q(i, jj)); };
        double v;
        if (j == 1)
          v = qy_edge(1, 2, 0, -1);
        else if (j == npy)
          v = qy_edge(npy - 1, npy - 2, npy, npy + 1);
        else if (j == 2) {
          const double g_in = FV3_M(dya, i, 2) / FV3_M(dya, i, 1);
          v = (3. * (g_in * q(i, 1) + q(i, 2)) - (g_in * qy_edge(1, 2, 0, -1) + qy_std(3))) / (2. + 2. * g_in);
        } else if (j == npy - 1) {
          const double g_in = FV3_M(dya, i, npy - 2) / FV3_M(dya, i, npy - 1);
          v = (3. * (q(i, npy - 2) + g_in * q(i, npy - 1)) - (g_in * qy_edge(npy - 1, npy - 2, npy, npy + 1) + qy_std(npy - 2))) / (2. + 2. * g_in);
        } else
          v = qy_std(j);
        view_A(g, s.qy[f])(i, j, k) = v;
      }
      // ---- corner values on the edges of the face and at its corners
      const VA o = view_A(g, s.out[f]);
      auto EXTRAP = [&](int n, double a, double b) { return a + s.cg.corner_f[n] * (a - b); };
      if ((i == 1 || i == npx) && j >= 2 && j <= npy - 1) {  // :143-150, :159-165
        const int ia = (i == 1) ? 0 : npx - 1, ib = ia + 1;
        auto q2 = [&](int jj) { return (q(ia, jj) * FV3_M(dxa, ib, jj) + q(ib, jj) * FV3_M(dxa, ia, jj)) / (FV3_M(dxa, ia, jj) + FV3_M(dxa, ib, jj)); };
        const double ew = (i == 1) ? s.cg.edge_w[j] : s.cg.edge_e[j];
        o(i, j, k) = ew * q2(j - 1) + (1. - ew) * q2(j);
      } else if ((j == 1 || j == npy) && i >= 2 && i <= npx - 1) {  // :193-199, :208-214
        const int ja = (j == 1) ? 0 : npy - 1, jb = ja + 1;
        auto q1 = [&](int ii) { return (q(ii, ja) * FV3_M(dya, ii, jb) + q(ii, jb) * FV3_M(dya, ii, ja)) / (FV3_M(dya, ii, ja) + FV3_M(dya, ii, jb)); };
        const double es = (j == 1) ? s.cg.edge_s[i] : s.cg.edge_n[i];
        o(i, j, k) = es * q1(i - 1) + (1. - es) * q1(i);
      } else if (i == 1 && j == 1) {  // :83-112
        o(1, 1, k) = (EXTRAP(0, q(1, 1), q(2, 2)) + EXTRAP(1, q(0, 1), q(-1, 2)) + EXTRAP(2, q(1, 0), q(2, -1))) * r3;
      } else if (i == npx && j == 1) {
        o(npx, 1, k) = (EXTRAP(3, q(npx - 1, 1), q(npx - 2, 2)) + EXTRAP(4, q(npx - 1, 0), q(npx - 2, -1)) + EXTRAP(5, q(npx, 1), q(npx + 1, 2))) * r3;
      } else if (i == npx && j == npy) {
        o(npx, npy, k) = (EXTRAP(6, q(npx - 1, npy - 1), q(npx - 2, npy - 2)) + EXTRAP(7, q(npx, npy - 1), q(npx + 1, npy - 2)) +
                          EXTRAP(8, q(npx - 1, npy), q(npx - 2, npy + 1))) * r3;
      } else if (i == 1 && j == npy) {
        o(1, npy, k) = (EXTRAP(9, q(1, npy - 1), q(2, npy - 2)) + EXTRAP(10, q(0, npy - 1), q(-1, npy - 2)) + EXTRAP(11, q(1, npy), q(2, npy + 1))) * r3;
      }
    }
  }
};

struct A2bCubedPb {
  A2bCubedState s;
  FV3_HD void operator()(int i, int j, int k) const {
    constexpr double a1 = 0.5625, a2 = -0.0625, c1 = 2. / 3., c2 = -1. / 6.;
    const Grid &g = s.g;
    const int npx = g.npx, npy = g.npy;
    for (int f = 0; f < s.nf; f++) {
      if (k >= s.nlev[f]) continue;
      const VA o = view_A(g, s.out[f]);
      if (k == 0 && ((s.override_mask >> f) & 1)) {
        o(i, j, k) = s.top[f];
        continue;
      }
      if (i < 2 || i > npx - 1 || j < 2 || j > npy - 1) continue;
      const CA qx = cview_A(g, s.qx[f]), qy = cview_A(g, s.qy[f]);
      auto qxx_std = [&](int jj) { return a2 * (qx(i, jj - 2, k) + qx(i, jj + 1, k)) + a1 * (qx(i, jj - 1, k) + qx(i, jj, k)); };
      auto qyy_std = [&](int ii) { return a2 * (qy(ii - 2, j, k) + qy(ii + 1, j, k)) + a1 * (qy(ii - 1, j, k) + qy(ii, j, k)); };
      double qxx, qyy;
      if (j == 2)
        qxx = c1 * (qx(i, 1, k) + qx(i, 2, k)) + c2 * (o(i, 1, k) + qxx_std(3));
      else if (j == npy - 1)
        qxx = c1 * (qx(i, npy - 2, k) + qx(i, npy - 1, k)) + c2 * (o(i, npy, k) + qxx_std(npy - 2));
      else
        qxx = qxx_std(j);
      if (i == 2)
        qyy = c1 * (qy(1, j, k) + qy(2, j, k)) + c2 * (o(1, j, k) + qyy_std(3));
      else if (i == npx - 1)
        qyy = c1 * (qy(npx - 2, j, k) + qy(npx - 1, j, k)) + c2 * (o(npx, j, k) + qyy_std(npx - 2));
      else
        qyy = qyy_std(i);
      o(i, j, k) = 0.5 * (qxx + qyy);
    }
  }
};

}  // namespace fv3
