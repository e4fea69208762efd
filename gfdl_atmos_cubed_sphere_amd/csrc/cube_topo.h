// cube_topo.h -- the halo topology of the six-face cubed sphere (one tile per face), host code of the library.
//
// What FMS mpp_update_domains / mpp_get_boundary do on the mosaic tools/fv_mp_mod.F90:498-546 defines (12 contacts; FMS itself is not
// part of the reference tree): a halo point of a face IS a point of the neighbour face, a vector component keeps its physical
// direction (DGRID_NE / CGRID_NE pairs swap members and change sign where the neighbour's axes are rotated).  Derived here from the
// geometry of the cube -- outward normal n and in-face axes ex, ey of every face, integer vectors -- not from a contact list; the
// oracle derives the same tables from the reference's contact list (oracle/fv_grid.c) and tests/test_grid_oracle.py holds the two
// equal row for row.  The tables drive the device gathers (six faces on one GPU) and the pack / unpack lists of the peer messages
// (one face per GPU, fv3_cube_halo_start).
#pragma once

#include <cstdlib>
#include <vector>

namespace fv3 {

struct V3 {
  int x, y, z;
};
static inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator*(int s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
static inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
static inline int dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// face 1 is centred on (lon 0, lat 0) with i eastward and j northward; the others follow from the contacts
struct Frame {
  V3 n, ex, ey;
};
static inline Frame cube_frame(int t) {
  const V3 X{1, 0, 0}, Y{0, 1, 0}, Z{0, 0, 1};
  switch (t) {
    case 0: return {X, Y, Z};
    case 1: return {Y, neg(X), Z};
    case 2: return {Z, neg(X), neg(Y)};
    case 3: return {neg(X), neg(Z), neg(Y)};
    case 4: return {neg(Y), neg(Z), X};
    default: return {neg(Z), Y, X};
  }
}

enum CubeKind { kCubeA = 0, kCubeB = 1, kCubeD = 2, kCubeC = 3, kCubeDedge = 4 };

struct CubeRow {
  long dst;   // flat Fortran-order index into the destination member array (incl. halo)
  int tile;   // source face 0..5
  int comp;   // 0: the source is the same member of the pair, 1: the other member
  long src;   // flat index into that source array
  int sign;   // +1 / -1 (vector updates; SCALAR_PAIR ignores it)
};

struct CubeTopo {
  int npx, ng, N;
  CubeTopo(int npx_, int ng_) : npx(npx_), ng(ng_), N(npx_ - 1) {}

  // member m of a kind: parity of the point (1 = cell centred in that direction) and the direction of the component (-1: scalar)
  static void member_of(int kind, int m, int &pa, int &pb, int &dir) {
    if (kind == kCubeA) { pa = 1; pb = 1; dir = -1; }
    else if (kind == kCubeB) { pa = 0; pb = 0; dir = -1; }
    else if (kind == kCubeD || kind == kCubeDedge) { if (m == 0) { pa = 1; pb = 0; dir = 0; } else { pa = 0; pb = 1; dir = 1; } }
    else { if (m == 0) { pa = 0; pb = 1; dir = 0; } else { pa = 1; pb = 0; dir = 1; } }
  }
  static int members(int kind) { return kind >= kCubeD ? 2 : 1; }
  static int member_at(int kind, int pa, int pb) {
    for (int m = 0; m < 2; m++) {
      int qa, qb, d;
      member_of(kind, m, qa, qb, d);
      if (qa == pa && qb == pb) return m;
    }
    std::abort();
  }
  V3 origin(int t) const {
    const Frame f = cube_frame(t);
    return N * (f.n - f.ex - f.ey);
  }
  long flat(int pa, int i, int j) const { return (long)(j - 1 + ng) * (N + 2 * ng + (1 - pa)) + (i - 1 + ng); }

  // the point p (3-D integer position in half-cell units) with outward direction nn of face t as seen from the face across that edge;
  // fold[d] = what the unit vector along the local axis d (0: x, 1: y) of face t becomes there
  void land(V3 p, V3 nn, const V3 fold[2], int &t2, int &a2, int &b2, int ax2[2], int sg2[2]) const {
    t2 = -1;
    for (int k = 0; k < 6; k++)
      if (same(cube_frame(k).n, nn)) t2 = k;
    if (t2 < 0) std::abort();
    const Frame f2 = cube_frame(t2);
    const V3 q = p - origin(t2);
    a2 = dot(q, f2.ex);
    b2 = dot(q, f2.ey);
    for (int d = 0; d < 2; d++) {
      const int sx = dot(fold[d], f2.ex), sy = dot(fold[d], f2.ey);
      if (sx != 0) { ax2[d] = 0; sg2[d] = sx; } else { ax2[d] = 1; sg2[d] = sy; }
    }
  }
  // doubled local coordinates (a, b) of face t, outside [0, 2N] in exactly one direction
  void map_point(int t, int a, int b, int &t2, int &a2, int &b2, int ax2[2], int sg2[2]) const {
    const int M = 2 * N;
    const Frame f = cube_frame(t);
    const V3 o = origin(t);
    V3 p, nn, fold[2];
    int d;
    if (a > M) { d = a - M; p = o + M * f.ex + b * f.ey; nn = f.ex; fold[0] = neg(f.n); fold[1] = f.ey; }
    else if (a < 0) { d = -a; p = o + b * f.ey; nn = neg(f.ex); fold[0] = f.n; fold[1] = f.ey; }
    else if (b > M) { d = b - M; p = o + a * f.ex + M * f.ey; nn = f.ey; fold[0] = f.ex; fold[1] = neg(f.n); }
    else { d = -b; p = o + a * f.ex; nn = neg(f.ey); fold[0] = f.ex; fold[1] = f.n; }
    p = p - d * f.n;
    land(p, nn, fold, t2, a2, b2, ax2, sg2);
  }

  // rows of the halo update of member m of `kind` on face t (kCubeDedge: mpp_get_boundary of (u, v): member 0 = u(i, npy) from
  // across the north edge, member 1 = v(npx, j) from across the east edge)
  std::vector<CubeRow> table(int kind, int m, int t) const {
    std::vector<CubeRow> rows;
    const int M = 2 * N;
    int pa, pb, dir;
    member_of(kind, m, pa, pb, dir);
    auto emit = [&](int i, int j, int t2, int a2, int b2, const int ax2[2], const int sg2[2]) {
      const int pa2 = ((a2 % 2) + 2) % 2, pb2 = ((b2 % 2) + 2) % 2;
      const int i2 = (a2 - pa2) / 2 + 1, j2 = (b2 - pb2) / 2 + 1;
      int m2 = m, sg = 1;
      if (dir >= 0) {
        m2 = member_at(kind, pa2, pb2);
        int qa, qb, d3;
        member_of(kind, m2, qa, qb, d3);
        if (d3 != ax2[dir]) std::abort();
        sg = sg2[dir];
      }
      rows.push_back({flat(pa, i, j), t2, m2 == m ? 0 : 1, flat(pa2, i2, j2), sg});
    };
    if (kind == kCubeDedge) {
      const Frame f = cube_frame(t);
      const V3 o = origin(t);
      for (int s = 1; s <= N; s++) {
        const int i = m == 0 ? s : N + 1, j = m == 0 ? N + 1 : s;
        const int a = m == 0 ? 2 * (s - 1) + 1 : M, b = m == 0 ? M : 2 * (s - 1) + 1;
        const V3 p = o + a * f.ex + b * f.ey;
        // the edge direction lies in both faces; the normal component is not exchanged
        V3 fold[2] = {f.ex, f.ey};
        int t2, a2, b2, ax2[2], sg2[2];
        land(p, m == 0 ? f.ey : f.ex, fold, t2, a2, b2, ax2, sg2);
        emit(i, j, t2, a2, b2, ax2, sg2);
      }
      return rows;
    }
    const int ei = 1 - pa, ej = 1 - pb;
    for (int j = 1 - ng; j <= N + ng + ej; j++)
      for (int i = 1 - ng; i <= N + ng + ei; i++) {
        const int a = 2 * (i - 1) + pa, b = 2 * (j - 1) + pb;
        const bool oa = a < 0 || a > M, ob = b < 0 || b > M;
        if (oa == ob) continue;   // interior / boundary point, or a corner region (no diagonal neighbour on the cube)
        int t2, a2, b2, ax2[2], sg2[2];
        map_point(t, a, b, t2, a2, b2, ax2, sg2);
        emit(i, j, t2, a2, b2, ax2, sg2);
      }
    return rows;
  }
};

}  // namespace fv3
