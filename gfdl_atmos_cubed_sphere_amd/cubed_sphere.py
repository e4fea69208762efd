"""The six-face cubed sphere: topology (which halo point is which point of which neighbour face, with the rotation of
vector components), the corner fills, and the gnomonic grid with the metric terms of ``fv_grid_type``.

Reference: the 12 contacts of the mosaic (tools/fv_mp_mod.F90:498-546), the halo semantics of FMS ``mpp_update_domains``
for positions CENTER / CORNER and vector grids DGRID_NE / CGRID_NE (FMS 2024.03 is not part of the reference tree, so
"parity unpinned" at that boundary -- SURVEY.md 8(c); here the rule is derived from the geometry of the cube: a halo
point IS a point of the neighbour face, a vector component keeps its physical direction), ``fill_corners``
(tools/fv_mp_mod.F90:944-1449), ``gnomonic_ed`` / ``symm_ed`` (model/fv_grid_utils.F90:1256-1351, :1530-1569),
``mirror_grid`` (tools/fv_grid_tools.F90:2625-2756), the metric computation of ``init_grid``
(tools/fv_grid_tools.F90:722-990) and ``grid_utils_init`` (model/fv_grid_utils.F90:84-790).

One tile per face (layout 1 x 1 per face -- one face per GPU, or all six faces on one GPU); every tile has four edges
and four corners.  Host-side set-up code (numpy); the tables built here drive the device gather / pack kernels.
"""
from __future__ import annotations

import numpy as np

from .grid import GridStruct
from .layout import Bounds

RADIUS = 6.3712e6      # FMS constants_mod
OMEGA = 7.2921e-5
BIG = float(__import__('os').environ.get('FV3_GRID_BIG', '1.0e30'))           # big_number (1e8 in fv_grid_utils.F90:56; larger here so that a read of an unset entry shows)
TINY = 1.0e-8          # tiny_number
NG = 3

# frames of the six faces on the cube: outward normal n, direction of increasing i (ex) and j (ey).  Face 1 is centred
# on (lon 0, lat 0) with i eastward and j northward; the others follow from the contacts (fv_mp_mod.F90:498-546):
# 1.E-2.W, 1.N-3.W(rev), 1.W-5.N(rev), 1.S-6.N, 2.N-3.S, 2.E-4.S(rev), 2.S-6.E(rev), 3.E-4.W, 3.N-5.W(rev), 4.N-5.S,
# 4.E-6.S(rev), 5.E-6.W
_X, _Y, _Z = np.array([1, 0, 0]), np.array([0, 1, 0]), np.array([0, 0, 1])
FRAMES = [(_X, _Y, _Z), (_Y, -_X, _Z), (_Z, -_X, -_Y), (-_X, -_Z, -_Y), (-_Y, -_Z, _X), (-_Z, _Y, _X)]

# point classes in doubled local coordinates (a, b) = (2(i-1) + pa, 2(j-1) + pb)
#   kind: (pa, pb, extra points in i, extra points in j)
_CLASS = {"A": (1, 1, 0, 0), "B": (0, 0, 1, 1), "U": (1, 0, 0, 1), "V": (0, 1, 1, 0)}


def _class_of(pa, pb):
    return {(1, 1): "A", (0, 0): "B", (1, 0): "U", (0, 1): "V"}[(pa % 2, pb % 2)]


class CubeTopology:
    """Index tables of the halo update of the six faces (one tile per face).

    ``table(kind)`` for kind in 'A' (cell centres), 'B' (corners), and the vector / scalar-pair grids 'D' (u on U points,
    v on V points: components ALONG the x / y cell edges), 'C' (uc on V points, vc on U points: components NORMAL to the
    edges), returns per destination array a dict of int arrays:
        dst (flat F-order index into the destination tile array), tile (source face 0..5), comp (0: the source is the
        same member of the pair, 1: the other member), src (flat F-order index into that source array), sign (+1/-1).
    Corner regions of the halo (both i and j outside 1..npx-1) have no source: the cube has no diagonal neighbour there;
    they are the business of fill_corners / copy_corners.
    """

    def __init__(self, npx: int, ng: int = NG):
        self.npx, self.ng, self.N = npx, ng, npx - 1
        self._cache = {}

    # shape of a tile array of a point class
    def shape(self, cls):
        n = self.N + 2 * self.ng
        _, _, ei, ej = _CLASS[cls]
        return (n + ei, n + ej)

    def _origin(self, t):
        n, ex, ey = FRAMES[t]
        return self.N * (n - ex - ey)

    def _map_point(self, t, a, b):
        """doubled local (a, b) of face t, outside [0, 2N] in exactly one direction -> (face', a', b', fold) where fold maps
        a local direction ('x' or 'y') to (axis', sign) on the neighbour"""
        M = 2 * self.N
        n, ex, ey = FRAMES[t]
        o = self._origin(t)
        if a > M:
            d, p, nn = a - M, o + M * ex + b * ey, ex
            fold = {"x": -n, "y": ey}
        elif a < 0:
            d, p, nn = -a, o + b * ey, -ex
            fold = {"x": n, "y": ey}
        elif b > M:
            d, p, nn = b - M, o + a * ex + M * ey, ey
            fold = {"x": ex, "y": -n}
        else:
            d, p, nn = -b, o + a * ex, -ey
            fold = {"x": ex, "y": n}
        p = p - d * n
        t2 = next(k for k in range(6) if np.array_equal(FRAMES[k][0], nn))
        n2, ex2, ey2 = FRAMES[t2]
        q = p - self._origin(t2)
        a2, b2 = int(q @ ex2), int(q @ ey2)
        assert 0 <= a2 <= M and 0 <= b2 <= M and int(q @ n2) == 0
        out = {}
        for k, v in fold.items():
            if abs(int(v @ ex2)) == 1:
                out[k] = ("x", int(v @ ex2))
            else:
                assert abs(int(v @ ey2)) == 1
                out[k] = ("y", int(v @ ey2))
        return t2, a2, b2, out

    def _flat(self, cls, i, j):
        """flat F-order index of Fortran (i, j) in a tile array of class cls"""
        ni = self.shape(cls)[0]
        return (j - 1 + self.ng) * ni + (i - 1 + self.ng)

    def table(self, kind: str):
        if kind in self._cache:
            return self._cache[kind]
        ng, N, M = self.ng, self.N, 2 * self.N
        # members of the pair: (class, direction of the component) ; scalars: one member without direction
        members = {"A": [("A", None)], "B": [("B", None)], "D": [("U", "x"), ("V", "y")], "C": [("V", "x"), ("U", "y")]}[kind]
        out = []
        for t in range(6):
            per_member = []
            for m, (cls, direc) in enumerate(members):
                pa, pb, ei, ej = _CLASS[cls]
                rows = []
                for j in range(1 - ng, N + ng + ej + 1):
                    for i in range(1 - ng, N + ng + ei + 1):
                        a, b = 2 * (i - 1) + pa, 2 * (j - 1) + pb
                        out_a, out_b = a < 0 or a > M, b < 0 or b > M
                        if out_a == out_b:          # interior / boundary point, or a corner region
                            continue
                        t2, a2, b2, fold = self._map_point(t, a, b)
                        cls2 = _class_of(a2, b2)
                        i2, j2 = (a2 - a2 % 2) // 2 + 1, (b2 - b2 % 2) // 2 + 1
                        sign, comp = 1, 0
                        if direc is not None:
                            axis, sign = fold[direc]
                            # which member of the pair lives on class cls2 with direction `axis`
                            m2 = next(k for k, (c, dd) in enumerate(members) if c == cls2 and dd == axis)
                            comp = 0 if m2 == m else 1
                        else:
                            assert cls2 == cls
                        rows.append((self._flat(cls, i, j), t2, comp, self._flat(cls2, i2, j2), sign))
                r = np.array(rows, dtype=np.int64)
                per_member.append(dict(dst=r[:, 0], tile=r[:, 1], comp=r[:, 2], src=r[:, 3], sign=r[:, 4]))
            out.append(per_member)
        self._cache[kind] = out
        return out

    def boundary_table(self):
        """mpp_get_boundary of (u, v) on the D grid (dyn_core.F90:1151-1163): u(i, npy) of a face takes the value the face
        across its north edge holds for the same edge, v(npx, j) the value of the face across the east edge.  Same row
        format as table('D'): [face][member] with member 0 = u rows (north edge), member 1 = v rows (east edge)."""
        if "Dedge" in self._cache:
            return self._cache["Dedge"]
        N, M = self.N, 2 * self.N
        members = [("U", "x"), ("V", "y")]
        out = []
        for t in range(6):
            n, ex, ey = FRAMES[t]
            o = self._origin(t)
            per = []
            for m, (cls, direc) in enumerate(members):
                rows = []
                for s in range(1, N + 1):
                    if m == 0:
                        i, j, a, b, nn = s, N + 1, 2 * (s - 1) + 1, M, ey
                    else:
                        i, j, a, b, nn = N + 1, s, M, 2 * (s - 1) + 1, ex
                    p = o + a * ex + b * ey
                    t2 = next(k for k in range(6) if np.array_equal(FRAMES[k][0], nn))
                    n2, ex2, ey2 = FRAMES[t2]
                    q = p - self._origin(t2)
                    a2, b2 = int(q @ ex2), int(q @ ey2)
                    d = ex if m == 0 else ey          # the edge direction lies in both faces
                    if abs(int(d @ ex2)) == 1:
                        axis, sign = "x", int(d @ ex2)
                    else:
                        axis, sign = "y", int(d @ ey2)
                    cls2 = _class_of(a2, b2)
                    m2 = next(k for k, (c, dd) in enumerate(members) if c == cls2 and dd == axis)
                    i2, j2 = (a2 - a2 % 2) // 2 + 1, (b2 - b2 % 2) // 2 + 1
                    rows.append((self._flat(cls, i, j), t2, 0 if m2 == m else 1, self._flat(cls2, i2, j2), sign))
                r = np.array(rows, dtype=np.int64)
                per.append(dict(dst=r[:, 0], tile=r[:, 1], comp=r[:, 2], src=r[:, 3], sign=r[:, 4]))
            out.append(per)
        self._cache["Dedge"] = out
        return out

    # ---- numpy halo update of all six faces (the single-process emulation of mpp_update_domains) -------------------------
    def update(self, kind: str, fields, vector: bool = True):
        """fields: for 'A' / 'B' a list of 6 arrays (ni, nj[, nk]); for 'D' / 'C' a pair (list of 6 first members, list of 6
        second members).  vector=False: SCALAR_PAIR (no sign change).  In place."""
        tab = self.boundary_table() if kind == "Dedge" else self.table(kind)
        pair = kind in ("D", "C", "Dedge")
        mem = fields if pair else (fields,)
        flat = [[np.reshape(x, (x.shape[0] * x.shape[1],) + x.shape[2:], order="F") for x in lst] for lst in mem]
        # gather every source value first (a halo never feeds a halo, but keep the update order-free)
        new = []
        for t in range(6):
            for m in range(len(mem)):
                tb = tab[t][m]
                vals = np.empty((tb["dst"].size,) + flat[m][t].shape[1:])
                for t2 in range(6):
                    for c in (0, 1):
                        sel = (tb["tile"] == t2) & (tb["comp"] == c)
                        if not sel.any():
                            continue
                        srcm = m if c == 0 else 1 - m
                        v = flat[srcm][t2][tb["src"][sel]]
                        if vector and pair:
                            s = tb["sign"][sel].astype(np.float64)
                            v = v * s.reshape((-1,) + (1,) * (v.ndim - 1))
                        vals[sel] = v
                new.append((t, m, tb["dst"], vals))
        for t, m, dst, vals in new:
            f = flat[m][t]
            f[dst] = vals
            mem[m][t][...] = np.reshape(f, mem[m][t].shape, order="F")


# ---- fill_corners (tools/fv_mp_mod.F90:944-1449): every tile owns all four corners here --------------------------------------
def _ix(a, ng):
    """Fortran-index view helper: returns f(i, j) -> numpy index tuple"""
    o = ng - 1
    return lambda i, j: (i + o, j + o)


def fill_corners_2d(q, npx, npy, fill: str, stagger: str, ng: int = NG):
    """fill_corners_2d (fv_mp_mod.F90:944-1016): q in place; fill 'x' (XDir) or 'y' (YDir); stagger 'B' (BGRID) or 'A' (AGRID)"""
    P = _ix(q, ng)
    for j in range(1, ng + 1):
        for i in range(1, ng + 1):
            if stagger == "B":
                if fill == "x":
                    q[P(1 - i, 1 - j)] = q[P(1 - j, i + 1)]
                    q[P(1 - i, npy + j)] = q[P(1 - j, npy - i)]
                    q[P(npx + i, 1 - j)] = q[P(npx + j, i + 1)]
                    q[P(npx + i, npy + j)] = q[P(npx + j, npy - i)]
                else:
                    q[P(1 - j, 1 - i)] = q[P(i + 1, 1 - j)]
                    q[P(1 - j, npy + i)] = q[P(i + 1, npy + j)]
                    q[P(npx + j, 1 - i)] = q[P(npx - i, 1 - j)]
                    q[P(npx + j, npy + i)] = q[P(npx - i, npy + j)]
            else:
                if fill == "x":
                    q[P(1 - i, 1 - j)] = q[P(1 - j, i)]
                    q[P(1 - i, npy - 1 + j)] = q[P(1 - j, npy - 1 - i + 1)]
                    q[P(npx - 1 + i, 1 - j)] = q[P(npx - 1 + j, i)]
                    q[P(npx - 1 + i, npy - 1 + j)] = q[P(npx - 1 + j, npy - 1 - i + 1)]
                else:
                    q[P(1 - j, 1 - i)] = q[P(i, 1 - j)]
                    q[P(1 - j, npy - 1 + i)] = q[P(i, npy - 1 + j)]
                    q[P(npx - 1 + j, 1 - i)] = q[P(npx - 1 - i + 1, 1 - j)]
                    q[P(npx - 1 + j, npy - 1 + i)] = q[P(npx - 1 - i + 1, npy - 1 + j)]


def fill_corners_xy(x, y, npx, npy, stagger: str, vector: bool = False, ng: int = NG):
    """fill_corners_xy_2d (fv_mp_mod.F90:1101-1130) -> fill_corners_dgrid / _cgrid / _agrid (:1290-1449).
    stagger 'D': x = (isd:ied, jsd:jed+1), y = (isd:ied+1, jsd:jed); 'C': x = (isd:ied+1, jsd:jed), y = (isd:ied, jsd:jed+1);
    'A': both cell centred."""
    s = -1.0 if vector else 1.0
    X, Y = _ix(x, ng), _ix(y, ng)
    for j in range(1, ng + 1):
        for i in range(1, ng + 1):
            if stagger == "D":
                x[X(1 - i, 1 - j)] = s * y[Y(1 - j, i)]
                x[X(1 - i, npy + j)] = y[Y(1 - j, npy - i)]
                x[X(npx - 1 + i, 1 - j)] = y[Y(npx + j, i)]
                x[X(npx - 1 + i, npy + j)] = s * y[Y(npx + j, npy - i)]
            elif stagger == "C":
                x[X(1 - i, 1 - j)] = y[Y(j, 1 - i)]
                x[X(1 - i, npy - 1 + j)] = s * y[Y(j, npy + i)]
                x[X(npx + i, 1 - j)] = s * y[Y(npx - j, 1 - i)]
                x[X(npx + i, npy - 1 + j)] = y[Y(npx - j, npy + i)]
            else:
                x[X(1 - i, 1 - j)] = s * y[Y(1 - j, i)]
                x[X(1 - i, npy - 1 + j)] = y[Y(1 - j, npy - 1 - i + 1)]
                x[X(npx - 1 + i, 1 - j)] = y[Y(npx - 1 + j, i)]
                x[X(npx - 1 + i, npy - 1 + j)] = s * y[Y(npx - 1 + j, npy - 1 - i + 1)]
    for j in range(1, ng + 1):
        for i in range(1, ng + 1):
            if stagger == "D":
                y[Y(1 - i, 1 - j)] = s * x[X(j, 1 - i)]
                y[Y(1 - i, npy - 1 + j)] = x[X(j, npy + i)]
                y[Y(npx + i, 1 - j)] = x[X(npx - j, 1 - i)]
                y[Y(npx + i, npy - 1 + j)] = s * x[X(npx - j, npy + i)]
            elif stagger == "C":
                y[Y(1 - i, 1 - j)] = x[X(1 - j, i)]
                y[Y(1 - i, npy + j)] = s * x[X(1 - j, npy - i)]
                y[Y(npx - 1 + i, 1 - j)] = s * x[X(npx + j, i)]
                y[Y(npx - 1 + i, npy + j)] = x[X(npx + j, npy - i)]
            else:
                y[Y(1 - j, 1 - i)] = s * x[X(i, 1 - j)]
                y[Y(1 - j, npy - 1 + i)] = x[X(i, npy - 1 + j)]
                y[Y(npx - 1 + j, 1 - i)] = x[X(npx - 1 - i + 1, 1 - j)]
                y[Y(npx - 1 + j, npy - 1 + i)] = s * x[X(npx - 1 - i + 1, npy - 1 + j)]


def fill_ghost(q, npx, npy, value, ng: int = NG):
    """fill_ghost (fv_grid_utils.F90): the four corner regions of a cell-centred array get `value`"""
    o = ng - 1
    n = q.shape[0]
    q[:ng, :ng] = value
    q[:ng, npy - 1 + o + 1:] = value
    q[npx - 1 + o + 1:, :ng] = value
    q[npx - 1 + o + 1:, npy - 1 + o + 1:] = value
    assert n == npx - 1 + 2 * ng


# ---- spherical geometry on unit vectors -------------------------------------------------------------------------------------------
def _unit(v):
    return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


def latlon_of(p):
    """cart_to_latlon (fv_grid_utils.F90:1682-1720): lon in [0, 2 pi), lat"""
    p = _unit(p)
    lon = np.where(np.abs(p[..., 0]) + np.abs(p[..., 1]) < 1e-10, 0.0, np.arctan2(p[..., 1], p[..., 0]))
    lon = np.where(lon < 0.0, 2.0 * np.pi + lon, lon)
    return lon, np.arcsin(p[..., 2])


def xyz_of(lon, lat):
    """latlon2xyz (fv_grid_utils.F90:1582-1608)"""
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1)


def gc_dist(lon1, lat1, lon2, lat2, radius=1.0):
    """great_circle_dist (fv_grid_utils.F90:1974-1996), haversine form"""
    beta = 2.0 * np.arcsin(np.sqrt(np.sin((lat1 - lat2) / 2.0) ** 2 + np.cos(lat1) * np.cos(lat2) * np.sin((lon1 - lon2) / 2.0) ** 2))
    return radius * beta


def _gcd3(p, q, radius=1.0):
    l1, t1 = latlon_of(p)
    l2, t2 = latlon_of(q)
    return gc_dist(l1, t1, l2, t2, radius)


def _mid(p, q):
    """mid_pt3_cart"""
    return _unit(p + q)


def _cross(a, b):
    return np.cross(a, b)


def _sph_angle(e1, e2, e3):
    """spherical_angle (fv_grid_utils.F90:2771-2828): angle at e1 between the arcs to e2 and e3"""
    p, q = _cross(e1, e2), _cross(e1, e3)
    ddd = np.sum(p * p, -1) * np.sum(q * q, -1)
    c = np.sum(p * q, -1) / np.sqrt(np.where(ddd > 0, ddd, 1.0))
    ang = np.arccos(np.clip(c, -1.0, 1.0))
    return np.where(ddd <= 0.0, 0.0, ang)


def _area(p1, p4, p2, p3, radius=1.0):
    """get_area(p1, p4, p2, p3) (fv_grid_utils.F90:2682-2723): p1 SW, p2 SE, p3 NE, p4 NW"""
    a = (_sph_angle(p1, p2, p4) + _sph_angle(p2, p3, p1) + _sph_angle(p3, p4, p2) + _sph_angle(p4, p3, p1))
    return (a - 2.0 * np.pi) * radius ** 2


def _area_tri(p1, p2, p3, radius=1.0):
    """get_area_tri (fv_grid_tools.F90:2358-2385): spherical excess of a triangle"""
    a = _sph_angle(p1, p2, p3) + _sph_angle(p2, p3, p1) + _sph_angle(p3, p1, p2)
    return (a - np.pi) * radius ** 2


def _cos_angle(p1, p2, p3):
    """cos_angle (fv_grid_utils.F90:2831-2875)"""
    p, q = _cross(p1, p2), _cross(p1, p3)
    ddd = np.sqrt(np.sum(p * p, -1) * np.sum(q * q, -1))
    return np.where(ddd > 0.0, np.sum(p * q, -1) / np.where(ddd > 0, ddd, 1.0), 1.0)


def gnomonic_ed_face(npx: int):
    """corner points (unit vectors, (npx, npx, 3)) of face 1: equidistant gnomonic grid (gnomonic_ed, fv_grid_utils.F90:1256-
    1351: equal angular spacing along the four edges, interior by the intersection of great circles = straight lines on the
    cube face), symmetrised about both mid-lines (symm_ed :1530-1569, mirror_grid fv_grid_tools.F90:2625-2667), centred on
    (lon 0, lat 0) with i eastward, j northward."""
    im = npx - 1
    rsq3 = 1.0 / np.sqrt(3.0)
    alpha = np.arcsin(rsq3)
    theta = -alpha + (2.0 * alpha / im) * np.arange(im + 1)
    z = np.sqrt(2.0) * rsq3 * np.tan(theta)          # edge points projected on the cube face x = 1/sqrt(3)
    z = 0.5 * (z - z[::-1])                          # exact antisymmetry about the mid-line (symm_ed)
    y = z
    p = np.empty((npx, npx, 3))
    p[..., 0] = rsq3
    p[..., 1] = y[:, None]
    p[..., 2] = z[None, :]
    return _unit(p)


class CubedSphere:
    """The six faces: corner / centre positions with halos and the gridstruct of every face."""

    def __init__(self, npx: int, ng: int = NG, radius: float = RADIUS, shift_fac: float = 18.0, omega: float = OMEGA):
        self.npx = self.npy = npx
        self.ng, self.radius, self.omega = ng, radius, omega
        self.topo = CubeTopology(npx, ng)
        N, o = npx - 1, ng - 1
        face1 = gnomonic_ed_face(npx)
        rot = -np.pi / shift_fac if shift_fac > 1e-4 else 0.0     # fv_grid_tools.F90:657-661: corner away from Japan
        cz, sz = np.cos(rot), np.sin(rot)
        Rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
        # corner points of every face incl. halo (the halo points ARE the neighbour's points)
        nb = N + 1 + 2 * ng
        self.grid3 = []
        for t in range(6):
            n, ex, ey = FRAMES[t]
            R = np.stack([n, ex, ey], axis=1).astype(np.float64)    # face-1 coordinates (x, y, z) -> n, ex, ey of face t
            g = np.full((nb, nb, 3), np.nan)
            g[ng:ng + npx, ng:ng + npx] = face1 @ R.T @ Rz.T
            self.grid3.append(g)
        for c in range(3):
            comp = [np.asfortranarray(g[..., c]) for g in self.grid3]
            self.topo.update("B", comp)
            for t in range(6):
                fill_corners_2d(comp[t], npx, npx, "x", "B", ng)                 # fv_grid_tools.F90:727-728
                self.grid3[t][..., c] = comp[t]
        self.grids = [self._face_metrics(t) for t in range(6)]
        self._exchange_metrics()

    # -- per-face metric terms ----------------------------------------------------------------------------------------------
    def _face_metrics(self, t):
        npx, ng, R = self.npx, self.ng, self.radius
        N = npx - 1
        g3 = self.grid3[t]                      # (nid+1, njd+1, 3), index = i - isd
        m = {}
        m["grid3"] = g3
        lon, lat = latlon_of(g3)
        m["grid"] = np.stack([lon, lat], axis=-1)
        # cell centres: cell_center2 (normalised sum of the four corners), compute domain; halo by exchange later
        a3 = _unit(g3[:-1, :-1] + g3[1:, :-1] + g3[:-1, 1:] + g3[1:, 1:])
        m["agrid3"] = a3
        # D-grid edge lengths on the compute domain (halo by exchange): dx (nid, njd+1), dy (nid+1, njd)
        m["dx"] = _gcd3(g3[1:, :], g3[:-1, :], R)
        m["dy"] = _gcd3(g3[:, 1:], g3[:, :-1], R)
        # cell widths through the mid points (fv_grid_tools.F90:813-823), all cells incl. halo
        m["dxa"] = _gcd3(_mid(g3[1:, :-1], g3[1:, 1:]), _mid(g3[:-1, :-1], g3[:-1, 1:]), R)
        m["dya"] = _gcd3(_mid(g3[:-1, 1:], g3[1:, 1:]), _mid(g3[:-1, :-1], g3[1:, :-1]), R)
        m["area"] = _area(g3[:-1, :-1], g3[:-1, 1:], g3[1:, :-1], g3[1:, 1:], R)
        return m

    def _exchange_metrics(self):
        npx, ng, R, topo = self.npx, self.ng, self.radius, self.topo
        N, o = npx - 1, ng - 1
        G = self.grids
        F = np.asfortranarray
        # agrid: exchange of the centres (they are the neighbour's centres), fill_corners lon XDir / lat YDir (:808-811)
        for c in range(3):
            comp = [F(g["agrid3"][..., c]) for g in G]
            topo.update("A", comp)
            for t in range(6):
                G[t]["agrid3"][..., c] = comp[t]
        for g in G:
            lon, lat = latlon_of(g["agrid3"])
            lon, lat = F(lon), F(lat)
            fill_corners_2d(lon, npx, npx, "x", "A", ng)
            fill_corners_2d(lat, npx, npx, "y", "A", ng)
            g["agrid"] = np.stack([lon, lat], axis=-1)
            g["agrid3"] = xyz_of(lon, lat)
        # dx, dy: halo = the neighbour's edge lengths (SCALAR_PAIR), corners by fill_corners DGRID (:779-783)
        dx, dy = [F(g["dx"]) for g in G], [F(g["dy"]) for g in G]
        topo.update("D", (dx, dy), vector=False)
        for t in range(6):
            fill_corners_xy(dx[t], dy[t], npx, npx, "D", ng=ng)
            G[t]["dx"], G[t]["dy"] = dx[t], dy[t]
        for t, g in enumerate(G):
            dxa, dya = F(g["dxa"]), F(g["dya"])
            fill_corners_xy(dxa, dya, npx, npx, "A", ng=ng)                       # :826-828
            g["dxa"], g["dya"] = dxa, dya
            a3 = g["agrid3"]
            nid = N + 2 * ng
            dxc = np.empty((nid + 1, nid))
            dxc[1:-1, :] = _gcd3(a3[1:, :], a3[:-1, :], R)                        # :836-842
            dxc[0, :], dxc[-1, :] = dxc[1, :], dxc[-2, :]
            dyc = np.empty((nid, nid + 1))
            dyc[:, 1:-1] = _gcd3(a3[:, 1:], a3[:, :-1], R)
            dyc[:, 0], dyc[:, -1] = dyc[:, 1], dyc[:, -2]
            # dual-cell areas (grid_area, fv_grid_tools.F90:2500-2580): corners i, j = 1..npx of the compute domain
            area_c = np.full((nid + 1, nid + 1), BIG)
            s = slice(ng, ng + npx)
            sm = slice(ng - 1, ng - 1 + npx)
            area_c[s, s] = _area(a3[sm, sm], a3[sm, s], a3[s, sm], a3[s, s], R)
            g3 = g["grid3"]
            P = lambda i, j: (i + o, j + o)          # noqa: E731
            # the three-cell corners of the cube: triangle of the three centres around it (:2540-2580)
            area_c[P(1, 1)] = _area_tri(a3[P(0, 1)], a3[P(1, 1)], a3[P(1, 0)], R)
            area_c[P(npx, 1)] = _area_tri(a3[P(npx, 1)], a3[P(npx - 1, 1)], a3[P(npx - 1, 0)], R)
            area_c[P(npx, npx)] = _area_tri(a3[P(npx - 1, npx)], a3[P(npx - 1, npx - 1)], a3[P(npx, npx - 1)], R)
            area_c[P(1, npx)] = _area_tri(a3[P(1, npx)], a3[P(1, npx - 1)], a3[P(0, npx - 1)], R)
            # face edges: twice the half cell on this face (fv_grid_tools.F90:871-936).  The reference's edge loops run over
            # js..je+1 / is..ie+1, i.e. they OVERWRITE the corner triangles of grid_area with twice the half cell too
            jj = np.arange(1, npx + 1)
            for i, inner in ((1, 1), (npx, npx - 1)):
                for j in jj:                 # do j = js, je+1: the corners included (the later j blocks overwrite them)
                    p1, p4 = _mid(g3[P(i, j - 1)], g3[P(i, j)]), _mid(g3[P(i, j)], g3[P(i, j + 1)])
                    p2, p3 = a3[P(inner, j - 1)], a3[P(inner, j)]
                    area_c[P(i, j)] = 2.0 * abs(_area(p1, p4, p2, p3, R))
                for j in range(1, npx):
                    dxc[P(i, j)] = 2.0 * _gcd3(_mid(g3[P(i, j)], g3[P(i, j + 1)]), a3[P(inner, j)], R)
            for j, inner in ((1, 1), (npx, npx - 1)):
                for i in jj:                 # do i = is, ie+1: last writer at the four cube corners (:905-912, :919-926)
                    p1, p2 = _mid(g3[P(i - 1, j)], g3[P(i, j)]), _mid(g3[P(i, j)], g3[P(i + 1, j)])
                    p3, p4 = a3[P(i, inner)], a3[P(i - 1, inner)]
                    area_c[P(i, j)] = 2.0 * abs(_area(p1, p4, p2, p3, R))
                for i in range(1, npx):
                    dyc[P(i, j)] = 2.0 * _gcd3(_mid(g3[P(i, j)], g3[P(i + 1, j)]), a3[P(i, inner)], R)
            g["dxc"], g["dyc"], g["area_c"] = F(dxc), F(dyc), F(area_c)
        dxc, dyc = [g["dxc"] for g in G], [g["dyc"] for g in G]
        topo.update("C", (dxc, dyc), vector=False)                                 # :939-943
        area = [F(g["area"]) for g in G]
        topo.update("A", area)
        area_c = [g["area_c"] for g in G]
        topo.update("B", area_c)
        for t, g in enumerate(G):
            fill_corners_xy(dxc[t], dyc[t], npx, npx, "C", ng=ng)
            fill_ghost(area[t], npx, npx, -BIG, ng)
            fill_corners_2d(area_c[t], npx, npx, "x", "B", ng)
            g["area"] = area[t]
        for t in range(6):
            self._angles(t)

    def _angles(self, t):
        """grid_utils_init (fv_grid_utils.F90:226-660): sin_sg / cos_sg at the 9 points of a cell, the edge / corner angle
        terms, divergence factors, Coriolis parameters, A->B edge interpolation factors"""
        npx, ng, R = self.npx, self.ng, self.radius
        g = self.grids[t]
        o = ng - 1
        g3, a3 = g["grid3"], g["agrid3"]
        nid = npx - 1 + 2 * ng
        sw, se, ne, nw = g3[:-1, :-1], g3[1:, :-1], g3[1:, 1:], g3[:-1, 1:]
        cos_sg = np.empty((nid, nid, 9))
        cos_sg[..., 5] = _cos_angle(sw, se, nw)                                    # 6: SW corner (:331-337)
        cos_sg[..., 6] = -_cos_angle(se, sw, ne)
        cos_sg[..., 7] = _cos_angle(ne, se, nw)
        cos_sg[..., 8] = -_cos_angle(nw, sw, ne)
        cos_sg[..., 0] = _cos_angle(_mid(sw, nw), a3, nw)                          # 1: west mid-point (:346-353)
        cos_sg[..., 1] = _cos_angle(_mid(sw, se), se, a3)
        cos_sg[..., 2] = _cos_angle(_mid(se, ne), a3, se)
        cos_sg[..., 3] = _cos_angle(_mid(nw, ne), nw, a3)
        # centre: ec1 . ec2 (get_center_vect: unit vectors along the mid-lines of the cell)
        pw, pe_, ps, pn = _mid(sw, nw), _mid(se, ne), _mid(sw, se), _mid(nw, ne)
        pc = _unit(sw + se + nw + ne)                                              # cell_center3
        ec1 = _unit(_cross(pc, _cross(pe_, pw)))                                   # get_center_vect :1762-1774
        ec2 = _unit(_cross(pc, _cross(pn, ps)))
        cos_sg[..., 4] = np.sum(ec1 * ec2, -1)
        g["ec1"], g["ec2"] = ec1, ec2
        sin_sg = np.minimum(1.0, np.sqrt(np.maximum(0.0, 1.0 - cos_sg ** 2)))
        P = lambda i, j: (i + o, j + o)      # noqa: E731
        # the first set of corner copies, sin_sg only, BEFORE the edge / corner averages (:363-394; the second set :573-612 below
        # follows fill_ghost).  It decides sina_u(1, j <= 0), sina_v(i <= 0, 1) ... in the halo rows next to the cube corners.
        for i in (-2, -1, 0):
            sin_sg[P(0, i) + (2,)] = sin_sg[P(i, 1) + (1,)]
            sin_sg[P(i, 0) + (3,)] = sin_sg[P(1, i) + (0,)]
        for i in range(npx, npx + 3):
            sin_sg[P(0, i) + (2,)] = sin_sg[P(npx - i, npx - 1) + (3,)]
        for i in (-2, -1, 0):
            sin_sg[P(i, npx) + (1,)] = sin_sg[P(1, npx + i) + (0,)]
        for j in (-2, -1, 0):
            sin_sg[P(npx, j) + (0,)] = sin_sg[P(npx - j, 1) + (1,)]
        for i in range(npx, npx + 3):
            sin_sg[P(i, 0) + (3,)] = sin_sg[P(npx - 1, npx - i) + (2,)]
        for i in range(npx, npx + 3):
            sin_sg[P(npx, i) + (0,)] = sin_sg[P(i, npx - 1) + (3,)]
            sin_sg[P(i, npx) + (1,)] = sin_sg[P(npx - 1, i) + (2,)]
        F = np.asfortranarray
        cosa = np.full((nid + 1, nid + 1), BIG)
        sina = np.full((nid + 1, nid + 1), BIG)
        s = slice(ng, ng + npx)
        sm = slice(ng - 1, ng - 1 + npx)
        cosa[s, s] = 0.5 * (cos_sg[sm, sm, 7] + cos_sg[s, s, 5])                   # :522-523
        sina[s, s] = 0.5 * (sin_sg[sm, sm, 7] + sin_sg[s, s, 5])
        cosa_u = np.full((nid + 1, nid), BIG)
        sina_u = np.full((nid + 1, nid), BIG)
        rsin_u = np.full((nid + 1, nid), BIG)
        cosa_u[1:-1, :] = 0.5 * (cos_sg[:-1, :, 2] + cos_sg[1:, :, 0])             # :533-539
        sina_u[1:-1, :] = 0.5 * (sin_sg[:-1, :, 2] + sin_sg[1:, :, 0])
        rsin_u[1:-1, :] = 1.0 / np.maximum(TINY, sina_u[1:-1, :] ** 2)
        cosa_v = np.full((nid, nid + 1), BIG)
        sina_v = np.full((nid, nid + 1), BIG)
        rsin_v = np.full((nid, nid + 1), BIG)
        cosa_v[:, 1:-1] = 0.5 * (cos_sg[:, :-1, 3] + cos_sg[:, 1:, 1])
        sina_v[:, 1:-1] = 0.5 * (sin_sg[:, :-1, 3] + sin_sg[:, 1:, 1])
        rsin_v[:, 1:-1] = 1.0 / np.maximum(TINY, sina_v[:, 1:-1] ** 2)
        cosa_s = F(cos_sg[..., 4].copy())
        rsin2 = F(1.0 / np.maximum(TINY, sin_sg[..., 4] ** 2))
        fill_ghost(cosa_s, npx, npx, BIG, ng)
        rsina = np.full((npx, npx), BIG)                                           # (is:ie+1, js:je+1), edges = big (:567-577)
        rsina[1:-1, 1:-1] = 1.0 / np.maximum(TINY, sina[s, s][1:-1, 1:-1] ** 2)
        for i in (1, npx):                                                         # :579-586 rsin_u = 1/sina_u on the face edges
            v = sina_u[i + o, :]
            rsin_u[i + o, :] = 1.0 / (np.sign(v) * np.maximum(TINY, np.abs(v)))
        for j in (1, npx):
            v = sina_v[:, j + o]
            rsin_v[:, j + o] = 1.0 / (np.sign(v) * np.maximum(TINY, np.abs(v)))
        for k in range(9):                                                         # :598-603
            sk, ck = F(sin_sg[..., k].copy()), F(cos_sg[..., k].copy())
            fill_ghost(sk, npx, npx, TINY, ng)
            fill_ghost(ck, npx, npx, BIG, ng)
            sin_sg[..., k], cos_sg[..., k] = sk, ck
        npy = npx
        for sg in (sin_sg, cos_sg):                                                # :608-645 (1-based plane n -> index n-1)
            for i in (0, -1, -2):
                sg[P(0, i) + (2,)] = sg[P(i, 1) + (1,)]
                sg[P(i, 0) + (3,)] = sg[P(1, i) + (0,)]
            for i in range(npy, npy + 3):
                sg[P(0, i) + (2,)] = sg[P(npy - i, npy - 1) + (3,)]
            for i in (0, -1, -2):
                sg[P(i, npy) + (1,)] = sg[P(1, npy - i) + (0,)]
            for j in (0, -1, -2):
                sg[P(npx, j) + (0,)] = sg[P(npx - j, 1) + (1,)]
            for i in range(npx, npx + 3):
                sg[P(i, 0) + (3,)] = sg[P(npx - 1, npx - i) + (2,)]
            for i in (0, 1, 2):
                sg[P(npx, npy + i) + (0,)] = sg[P(npx + i, npy - 1) + (3,)]
                sg[P(npx + i, npy) + (1,)] = sg[P(npx - 1, npy + i) + (2,)]
        dx, dy, dxc, dyc = g["dx"], g["dy"], g["dxc"], g["dyc"]
        divg_u = sina_v * dyc / dx                                                 # :676-700
        del6_u = sina_v * dx / dyc
        for j in (1, npy):
            sv = 0.5 * (sin_sg[:, j + o, 1] + sin_sg[:, j - 1 + o, 3])
            divg_u[:, j + o] = sv * dyc[:, j + o] / dx[:, j + o]
            del6_u[:, j + o] = sv * dx[:, j + o] / dyc[:, j + o]
        divg_v = sina_u * dxc / dy
        del6_v = sina_u * dy / dxc
        for i in (1, npx):
            su = 0.5 * (sin_sg[i + o, :, 0] + sin_sg[i - 1 + o, :, 2])
            divg_v[i + o, :] = su * dxc[i + o, :] / dy[i + o, :]
            del6_v[i + o, :] = su * dy[i + o, :] / dxc[i + o, :]
        g.update(cos_sg=cos_sg, sin_sg=sin_sg, cosa=F(cosa), sina=F(sina), cosa_u=F(cosa_u), sina_u=F(sina_u), rsin_u=F(rsin_u),
                 cosa_v=F(cosa_v), sina_v=F(sina_v), rsin_v=F(rsin_v), cosa_s=cosa_s, rsin2=rsin2, rsina=F(rsina),
                 divg_u=F(divg_u), del6_u=F(del6_u), divg_v=F(divg_v), del6_v=F(del6_v))
        # Coriolis (init_case / fv_grid_tools: f0 = 2 omega sin(lat) on agrid, fC on grid)
        g["f0"] = F(2.0 * self.omega * np.sin(g["agrid"][..., 1]))
        g["fC"] = F(2.0 * self.omega * np.sin(g["grid"][..., 1]))
        # edge_factors (fv_grid_utils.F90:1121-1230): A -> B interpolation weights on the four face edges
        lonlat = lambda p: latlon_of(p)     # noqa: E731
        edge = {k: np.full(npx, BIG) for k in "wesn"}
        for name, i in (("w", 1), ("e", npx)):
            py = _mid(a3[i - 1 + o, :], a3[i + o, :])                               # index by j + o
            for j in range(2, npx):
                gp = g3[P(i, j)]
                d1, d2 = _gcd3(py[j - 1 + o], gp), _gcd3(py[j + o], gp)
                edge[name][j - 1] = d2 / (d1 + d2)
        for name, j in (("s", 1), ("n", npx)):
            px = _mid(a3[:, j - 1 + o], a3[:, j + o])
            for i in range(2, npx):
                gp = g3[P(i, j)]
                d1, d2 = _gcd3(px[i - 1 + o], gp), _gcd3(px[i + o], gp)
                edge[name][i - 1] = d2 / (d1 + d2)
        g["edge_w"], g["edge_e"], g["edge_s"], g["edge_n"] = edge["w"], edge["e"], edge["s"], edge["n"]
        del lonlat
        # extrap_corner factors x1 / (x2 - x1) of a2b_ord4 (a2b_edge.F90:83-112, :452-462): corners sw, se, ne, nw, the
        # three (inner, outer) centre pairs in the reference's order
        def fac(p0, pa, pb):
            x1, x2 = _gcd3(a3[P(*pa)], g3[P(*p0)]), _gcd3(a3[P(*pb)], g3[P(*p0)])
            return x1 / (x2 - x1)
        n = npx
        g["corner_f"] = np.array([
            [fac((1, 1), (1, 1), (2, 2)), fac((1, 1), (0, 1), (-1, 2)), fac((1, 1), (1, 0), (2, -1))],
            [fac((n, 1), (n - 1, 1), (n - 2, 2)), fac((n, 1), (n - 1, 0), (n - 2, -1)), fac((n, 1), (n, 1), (n + 1, 2))],
            [fac((n, n), (n - 1, n - 1), (n - 2, n - 2)), fac((n, n), (n, n - 1), (n + 1, n - 2)), fac((n, n), (n - 1, n), (n - 2, n + 1))],
            [fac((1, n), (1, n - 1), (2, n - 2)), fac((1, n), (0, n - 1), (-1, n - 2)), fac((1, n), (1, n), (2, n + 1))]])

    def finalize_pairs(self):
        """divg_v / divg_u and del6_v / del6_u get their halos from the neighbours (fv_grid_utils.F90:717-720) and the global
        minima da_min, da_min_c are formed"""
        G = self.grids
        for a, b in (("divg_v", "divg_u"), ("del6_v", "del6_u")):
            x, y = [g[a] for g in G], [g[b] for g in G]
            self.topo.update("C", (x, y), vector=False)
        ng, npx = self.ng, self.npx
        s = slice(ng, ng + npx - 1)
        self.da_min = min(float(g["area"][s, s].min()) for g in G)
        self.da_max = max(float(g["area"][s, s].max()) for g in G)
        sc = slice(ng, ng + npx - 1)       # global_mx_c over (is:ie, js:je) of area_c
        self.da_min_c = min(float(g["area_c"][sc, sc].min()) for g in G)
        self.da_max_c = max(float(g["area_c"][sc, sc].max()) for g in G)

    def gridstruct(self, t: int) -> GridStruct:
        """the fv_grid_type of face t (grid_type = 0, one tile per face: all four corner flags set)"""
        if not hasattr(self, "da_min"):
            self.finalize_pairs()
        g = self.grids[t]
        npx = self.npx
        bd = Bounds(1, npx - 1, 1, npx - 1, ng=self.ng)
        F = np.asfortranarray
        gs = GridStruct(bd=bd, npx=npx, npy=npx, grid_type=0)
        m = gs.m
        for n in ("area", "dxa", "dya", "cosa_s", "rsin2", "f0", "dx", "dy", "dxc", "dyc", "cosa_u", "sina_u", "rsin_u", "cosa_v",
                  "sina_v", "rsin_v", "divg_u", "del6_u", "divg_v", "del6_v", "fC", "cosa", "sina", "rsina"):
            m[n] = F(g[n])
        for n, src in (("rarea", "area"), ("rdxa", "dxa"), ("rdya", "dya"), ("rdx", "dx"), ("rdy", "dy"), ("rdxc", "dxc"),
                       ("rdyc", "dyc"), ("rarea_c", "area_c")):
            m[n] = F(1.0 / g[src])
        m["sin_sg"], m["cos_sg"] = F(g["sin_sg"]), F(g["cos_sg"])
        gs.da_min, gs.da_min_c = self.da_min, self.da_min_c
        gs.sw_corner = gs.se_corner = gs.ne_corner = gs.nw_corner = True
        for n in ("edge_w", "edge_e", "edge_s", "edge_n", "corner_f"):
            m[n] = g[n]
        m["grid"], m["agrid"] = F(g["grid"]), F(g["agrid"])
        # init_cubed_to_latlon (fv_grid_utils.F90:2255-2315): the matrix from the cell-centre covariant winds to (east, north)
        lon, lat = g["agrid"][..., 0], g["agrid"][..., 1]
        vlon = np.stack([-np.sin(lon), np.cos(lon), np.zeros_like(lon)], axis=-1)                       # unit_vect_latlon :2220-2243
        vlat = np.stack([-np.sin(lat) * np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat)], axis=-1)
        z11, z12 = np.sum(g["ec1"] * vlon, -1), np.sum(g["ec1"] * vlat, -1)
        z21, z22 = np.sum(g["ec2"] * vlon, -1), np.sum(g["ec2"] * vlat, -1)
        s5 = g["sin_sg"][..., 4]
        m["a11"], m["a12"] = F(0.5 * z22 / s5), F(-0.5 * z12 / s5)
        m["a21"], m["a22"] = F(-0.5 * z21 / s5), F(0.5 * z11 / s5)
        # unit vectors of adv_pe (dyn_core.F90:1529): ec1, ec2 at the centres (A layout x 3, get_center_vect) and the edge normals
        # en1 (is:ie, js:je+1) = grid3(i,j) x grid3(i+1,j), en2 (is:ie+1, js:je) = grid3(i,j+1) x grid3(i,j) (fv_grid_utils.F90:632-643)
        g3 = g["grid3"]
        o_, n_ = self.ng, self.npx - 1
        c0 = g3[o_:o_ + n_ + 1, o_:o_ + n_ + 1]
        m["ec1"], m["ec2"] = F(g["ec1"]), F(g["ec2"])
        m["en1"] = F(_unit(_cross(c0[:-1, :], c0[1:, :])))
        m["en2"] = F(_unit(_cross(c0[:, 1:], c0[:, :-1])))
        gs.tile = t
        return gs
