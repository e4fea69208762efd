"""Field layout contract of the FV3 dynamical core (host-side mirror).

Mirrors the index contract of ``fv_grid_bounds_type`` (model/fv_arrays.F90:1192-1200) and the
field shapes of ``allocate_fv_atmos_type`` (model/fv_arrays.F90:1521-1563,1614-1631) and of the
``dyn_core`` work arrays (model/dyn_core.F90:256-283).  Every array is Fortran column-major
(i fastest, then j, then k) with the reference's exact lower/upper bounds, so a buffer made here
can be handed unchanged to the C-ABI (include/fv3_mi355x.h) or to a Fortran caller.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# stagger kind -> (i_lo, i_hi, j_lo, j_hi) in terms of bounds attribute names + offsets
_KINDS = {
    "A": (("isd", 0), ("ied", 0), ("jsd", 0), ("jed", 0)),    # cell centres, full halo
    "U": (("isd", 0), ("ied", 0), ("jsd", 0), ("jed", 1)),    # D-grid u / C-grid vc
    "V": (("isd", 0), ("ied", 1), ("jsd", 0), ("jed", 0)),    # D-grid v / C-grid uc
    "B": (("isd", 0), ("ied", 1), ("jsd", 0), ("jed", 1)),    # corners, full halo
    "CX": (("is_", 0), ("ie", 1), ("jsd", 0), ("jed", 0)),    # crx, xfx, cx
    "CY": (("isd", 0), ("ied", 0), ("js", 0), ("je", 1)),     # cry, yfx, cy
    "FX": (("is_", 0), ("ie", 1), ("js", 0), ("je", 0)),      # fx, mfx
    "FY": (("is_", 0), ("ie", 0), ("js", 0), ("je", 1)),      # fy, mfy
    "CC": (("is_", 0), ("ie", 0), ("js", 0), ("je", 0)),      # compute cells (delz, pkz, heat_s)
    "BC": (("is_", 0), ("ie", 1), ("js", 0), ("je", 1)),      # compute corners (rsina, ke)
    "RX": (("is_", 0), ("ie", 0), ("jsd", 0), ("jed", 0)),    # ra_x
    "RY": (("isd", 0), ("ied", 0), ("js", 0), ("je", 0)),     # ra_y
}


@dataclass(frozen=True)
class Bounds:
    """fv_grid_bounds_type: compute domain is:ie x js:je, data domain = +- ng."""

    is_: int
    ie: int
    js: int
    je: int
    ng: int = 3

    @property
    def isd(self) -> int:
        return self.is_ - self.ng

    @property
    def ied(self) -> int:
        return self.ie + self.ng

    @property
    def jsd(self) -> int:
        return self.js - self.ng

    @property
    def jed(self) -> int:
        return self.je + self.ng

    @property
    def nx(self) -> int:
        return self.ie - self.is_ + 1

    @property
    def ny(self) -> int:
        return self.je - self.js + 1

    def limits(self, kind: str):
        (a, da), (b, db), (c, dc), (d, dd) = _KINDS[kind]
        return (getattr(self, a) + da, getattr(self, b) + db, getattr(self, c) + dc, getattr(self, d) + dd)

    def shape(self, kind: str, nk: int | None = None):
        ilo, ihi, jlo, jhi = self.limits(kind)
        s = (ihi - ilo + 1, jhi - jlo + 1)
        return s if nk is None else s + (nk,)

    def zeros(self, kind: str, nk: int | None = None) -> np.ndarray:
        return np.zeros(self.shape(kind, nk), dtype=np.float64, order="F")

    def full(self, kind: str, value: float, nk: int | None = None) -> np.ndarray:
        return np.full(self.shape(kind, nk), value, dtype=np.float64, order="F")

    def view(self, arr: np.ndarray, kind: str, i0: int, i1: int, j0: int, j1: int) -> np.ndarray:
        """Slice ``arr`` (of stagger ``kind``) on the Fortran index range i0:i1, j0:j1 (inclusive)."""
        ilo, _, jlo, _ = self.limits(kind)
        return arr[i0 - ilo : i1 - ilo + 1, j0 - jlo : j1 - jlo + 1]


def periodic_fill(b: Bounds, arr: np.ndarray, kind: str, fill_edge: bool = False) -> None:
    """Fill the halo of a doubly periodic single-tile field in place (the FMS periodic contacts of
    tools/fv_mp_mod.F90:473-483 applied to one rank).  For north-east staggered kinds the edge
    row/column (index je+1 / ie+1) is part of the compute domain and is left alone unless
    ``fill_edge`` (then it is copied from index js / is, which makes a synthetic input consistent)."""
    ilo, ihi, jlo, jhi = b.limits(kind)
    nx, ny = b.nx, b.ny
    ii = np.arange(ilo, ihi + 1)
    jj = np.arange(jlo, jhi + 1)
    ei = b.ie + (1 if (kind in ("V", "B") and not fill_edge) else 0)
    ej = b.je + (1 if (kind in ("U", "B") and not fill_edge) else 0)
    src_i = np.where(ii < b.is_, ii + nx, np.where(ii > ei, ii - nx, ii)) - ilo
    src_j = np.where(jj < b.js, jj + ny, np.where(jj > ej, jj - ny, jj)) - jlo
    arr[...] = arr[np.ix_(src_i, src_j)] if arr.ndim == 2 else arr[src_i][:, src_j]
