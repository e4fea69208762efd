"""Host-side ``gridstruct``: the metric terms the dyn_core hot path reads.

Member names and shapes follow ``fv_grid_type`` (model/fv_arrays.F90:75-205, allocation shapes
:1749-1881).  ``doubly_periodic`` reproduces what the reference sets for ``grid_type=4``
(tools/fv_grid_tools.F90:1202-1221 ``setup_cartesian``; model/fv_grid_utils.F90:426-437,614-626,
656-665,680-683; f-plane tools/test_cases.F90:4682-4684) -- SURVEY.md appendix A.
``perturbed`` produces smooth array-valued (non-constant, non-orthogonal) metrics so that the
parity tests exercise every metric term of the kernels, not only the constant ones.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .layout import Bounds

OMEGA = 7.292e-5  # constants_mod omega (FMS), used only for the f-plane value

# name -> stagger kind (see layout._KINDS)
METRIC_KINDS = {
    "area": "A", "rarea": "A", "dxa": "A", "dya": "A", "rdxa": "A", "rdya": "A",
    "cosa_s": "A", "rsin2": "A", "f0": "A",
    "dx": "U", "rdx": "U", "dyc": "U", "rdyc": "U", "cosa_v": "U", "sina_v": "U", "rsin_v": "U",
    "divg_u": "U", "del6_u": "U",
    "dy": "V", "rdy": "V", "dxc": "V", "rdxc": "V", "cosa_u": "V", "sina_u": "V", "rsin_u": "V",
    "divg_v": "V", "del6_v": "V",
    "rarea_c": "B", "fC": "B", "cosa": "B", "sina": "B",
    "rsina": "BC",
}


@dataclass
class GridStruct:
    bd: Bounds
    npx: int
    npy: int
    grid_type: int = 4
    da_min: float = 0.0
    da_min_c: float = 0.0
    bounded_domain: bool = False
    stretched_grid: bool = False
    sw_corner: bool = False
    se_corner: bool = False
    ne_corner: bool = False
    nw_corner: bool = False
    # flagstruct members read by the kernels
    lim_fac: float = 1.0
    do_diss_est: bool = False
    prevent_diss_cooling: bool = True
    do_f3d: bool = False
    m: dict = field(default_factory=dict)  # metric arrays incl. sin_sg, cos_sg (isd:ied,jsd:jed,9)

    def __getattr__(self, name):
        m = self.__dict__.get("m", {})
        if name in m:
            return m[name]
        raise AttributeError(name)


def doubly_periodic(bd: Bounds, npx: int, npy: int, dx_const: float = 1000.0, dy_const: float = 1000.0,
                    deglat: float = 15.0) -> GridStruct:
    g = GridStruct(bd=bd, npx=npx, npy=npy, grid_type=4)
    m = g.m
    for name, kind in METRIC_KINDS.items():
        m[name] = bd.zeros(kind)
    for n in ("dx", "dxc", "dxa"):
        m[n][...] = dx_const
    for n in ("dy", "dyc", "dya"):
        m[n][...] = dy_const
    for n in ("rdx", "rdxc", "rdxa"):
        m[n][...] = 1.0 / dx_const
    for n in ("rdy", "rdyc", "rdya"):
        m[n][...] = 1.0 / dy_const
    m["area"][...] = dx_const * dy_const
    m["rarea"][...] = 1.0 / (dx_const * dy_const)
    m["rarea_c"][...] = 1.0 / (dx_const * dy_const)
    for n in ("sina", "rsina", "rsin2", "sina_u", "sina_v", "rsin_u", "rsin_v"):
        m[n][...] = 1.0
    # cosa*, cosa_s stay 0
    m["divg_u"][...] = m["sina_v"] * m["dyc"] / m["dx"]
    m["del6_u"][...] = m["sina_v"] * m["dx"] / m["dyc"]
    m["divg_v"][...] = m["sina_u"] * m["dxc"] / m["dy"]
    m["del6_v"][...] = m["sina_u"] * m["dy"] / m["dxc"]
    f = 2.0 * OMEGA * np.sin(np.deg2rad(deglat))
    m["f0"][...] = f
    m["fC"][...] = f
    m["sin_sg"] = np.ones(bd.shape("A", 9), order="F")
    m["cos_sg"] = np.zeros(bd.shape("A", 9), order="F")
    g.da_min = g.da_min_c = dx_const * dy_const
    return g


ANGLE_TERMS = ("cosa", "cosa_s", "cosa_u", "cosa_v", "sina", "rsina", "rsin2", "sina_u", "sina_v", "rsin_u", "rsin_v")


def perturbed(g: GridStruct, seed: int = 7, amp: float = 0.05, ortho: bool = False) -> GridStruct:
    """Smoothly perturb every metric term of a doubly periodic gridstruct (test helper).  The
    result is not a geometrically consistent grid; it is a set of positive, smooth arrays that
    makes every metric read by the kernels matter in a parity comparison.  ortho=True leaves the
    angle terms (cosa* = 0, sin* = 1) exact: an orthogonal grid with varying lengths and areas."""
    bd = g.bd
    rng = np.random.default_rng(seed)
    out = GridStruct(bd=bd, npx=g.npx, npy=g.npy, grid_type=g.grid_type, da_min=g.da_min,
                     da_min_c=g.da_min_c * (1.0 + 0.01), lim_fac=g.lim_fac, do_diss_est=g.do_diss_est,
                     prevent_diss_cooling=g.prevent_diss_cooling)

    def smooth(shape):
        ni, nj = shape[0], shape[1]
        x = np.arange(ni)[:, None] / max(ni, 1)
        y = np.arange(nj)[None, :] / max(nj, 1)
        ph = rng.uniform(0, 2 * np.pi, 4)
        s = (np.sin(2 * np.pi * x + ph[0]) * np.cos(2 * np.pi * y + ph[1])
             + 0.5 * np.sin(4 * np.pi * x + ph[2]) * np.sin(2 * np.pi * y + ph[3]))
        return s

    m = out.m
    for name, kind in METRIC_KINDS.items():
        base = g.m[name]
        if ortho and name in ANGLE_TERMS:
            smooth(base.shape)  # keep the random stream of the other terms
            m[name] = np.asfortranarray(base.copy())
        elif name in ("cosa", "cosa_s", "cosa_u", "cosa_v"):
            m[name] = np.asfortranarray(amp * smooth(base.shape))
        else:
            m[name] = np.asfortranarray(base * (1.0 + amp * smooth(base.shape)))
    sg = np.empty(bd.shape("A", 9), order="F")
    cg = np.empty(bd.shape("A", 9), order="F")
    for n in range(9):
        sg[:, :, n] = 1.0 - 0.5 * amp * (1.0 + smooth(sg.shape))  # in (1-amp*..., 1]
        cg[:, :, n] = amp * smooth(cg.shape)
    if ortho:
        sg[...] = 1.0
        cg[...] = 0.0
    m["sin_sg"], m["cos_sg"] = sg, cg
    return out
