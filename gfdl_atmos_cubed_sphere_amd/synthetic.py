"""Seeded synthetic states of the hot path's inputs (smooth sin/cos winds + noise, a nearly
hydrostatic column state), on the reference's field layout with valid periodic halos.  Used by bench.py, tools/ and the
parity tests -- host-side input generation only, no reference arithmetic."""
from __future__ import annotations

import numpy as np

from .layout import Bounds, periodic_fill
from .lib import GRAV, KAPPA, RDGAS

SEED = 20260928

# c_sw's outputs and their stagger (model/sw_core.F90:79-81)
CSW_OUT = (("delpc", "A"), ("ptc", "A"), ("wc", "A"), ("uc", "V"), ("vc", "U"), ("ua", "A"), ("va", "A"),
           ("ut", "A"), ("vt", "A"), ("divg_d", "B"))
# d_sw's scalar arguments at the reference defaults (hord 10/10/10/10, hord_tr 8, d4_bg 0.16)
DSW_PAR = dict(dt=6.0, hord_tr=8, hord_mt=10, hord_vt=10, hord_tm=10, hord_dp=10, dddmp=0.0, d4_bg=0.16, kgb=0.0)


def smooth_state(bd: Bounds, npz: int, seed: int = SEED, hydrostatic: bool = False, noise: float = 1.0):
    """Doubly periodic prognostic state with valid halos: u,v (D-grid), delp, pt, w."""
    rng = np.random.default_rng(seed)
    nx, ny = bd.nx, bd.ny

    def xy(kind):
        ilo, ihi, jlo, jhi = bd.limits(kind)
        x = (np.arange(ilo, ihi + 1) - bd.is_) / nx
        y = (np.arange(jlo, jhi + 1) - bd.js) / ny
        return x[:, None, None], y[None, :, None]

    kk = np.arange(npz)[None, None, :] / max(npz, 1)
    f = {}
    x, y = xy("U")
    f["u"] = 10.0 + 5.0 * np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 2.0 * np.cos(2 * np.pi * (x + kk))
    x, y = xy("V")
    f["v"] = -3.0 + 5.0 * np.cos(2 * np.pi * x) * np.sin(4 * np.pi * y) + 2.0 * np.sin(2 * np.pi * (y - kk))
    x, y = xy("A")
    f["delp"] = 800.0 * (1.0 + 0.2 * np.sin(2 * np.pi * x) * np.sin(2 * np.pi * y) + 0.1 * kk)
    f["pt"] = 300.0 + 10.0 * np.cos(2 * np.pi * x) * np.cos(4 * np.pi * y) + 20.0 * kk
    f["w"] = 0.5 * np.sin(4 * np.pi * x) * np.cos(2 * np.pi * y) + 0.0 * kk
    out = {}
    for n, kind in (("u", "U"), ("v", "V"), ("delp", "A"), ("pt", "A"), ("w", "A")):
        a = np.asfortranarray(np.broadcast_to(f[n], bd.shape(kind, npz)).copy())
        scale = {"u": 1.0, "v": 1.0, "delp": 8.0, "pt": 1.0, "w": 0.1}[n]
        a += noise * scale * rng.uniform(-1.0, 1.0, a.shape)
        for k in range(npz):
            periodic_fill(bd, a[:, :, k], kind, fill_edge=True)
        out[n] = a
    if hydrostatic:
        out.pop("w")
    return out


PTOP = 300.0


def nh_state(bd: Bounds, km: int, seed: int = 11, pert: float = 0.02):
    """A nearly hydrostatic column state on the reference layout with valid halos."""
    rng = np.random.default_rng(seed)
    sig = np.linspace(0.0, 1.0, km + 1) ** 1.5
    shapeA = bd.shape("A")
    ps = 1.0e5 * (1.0 + 0.01 * rng.uniform(-1, 1, shapeA))
    periodic_fill(bd, ps, "A")
    pe = PTOP + (ps[:, :, None] - PTOP) * sig[None, None, :]
    delp = np.asfortranarray(np.diff(pe, axis=2))
    pm = delp / np.log(pe[:, :, 1:] / pe[:, :, :-1])
    T = 300.0 - 60.0 * (1.0 - sig[None, None, 1:]) + 2.0 * rng.uniform(-1, 1, delp.shape)
    pt = np.asfortranarray(T * pm ** (-KAPPA))
    dz = -delp / GRAV * RDGAS * pt * pm ** (KAPPA - 1.0) * (1.0 + pert * rng.uniform(-1, 1, delp.shape))
    zs = np.asfortranarray(50.0 * rng.uniform(0, 1, shapeA))
    periodic_fill(bd, zs, "A")
    for k in range(km):
        periodic_fill(bd, pt[:, :, k], "A")
        periodic_fill(bd, dz[:, :, k], "A")
    zh = np.zeros(bd.shape("A", km + 1), order="F")
    zh[:, :, km] = zs
    for k in range(km - 1, -1, -1):
        zh[:, :, k] = zh[:, :, k + 1] - dz[:, :, k]
    w = np.asfortranarray(0.5 * rng.uniform(-1, 1, delp.shape))
    for k in range(km):
        periodic_fill(bd, w[:, :, k], "A")
    dp0 = np.diff(PTOP + (1.0e5 - PTOP) * sig)
    return dict(delp=delp, pt=pt, w=w, zh=zh, zs=zs, dp0=dp0)


def balanced_nh_state(bd: Bounds, npz: int, seed: int = 21):
    """u, v, w, delp, pt, delz, phis of a nearly hydrostatic atmosphere + the reference-pressure thicknesses dp0"""
    s = nh_state(bd, npz, seed=seed, pert=0.005)
    w = smooth_state(bd, npz, noise=0.05)
    delz = np.asfortranarray(np.diff(s["zh"], axis=2)[bd.ng:bd.ng + bd.nx, bd.ng:bd.ng + bd.ny, :])  # zh(k+1)-zh(k) < 0
    return dict(u=w["u"], v=w["v"], w=np.asfortranarray(0.2 * w["w"]), delp=s["delp"], pt=s["pt"], delz=delz,
                phis=np.asfortranarray(s["zs"] * GRAV)), s["dp0"]
