"""Seeded synthetic states for the parity tests (PCG64 seed 20260928,
smooth sin/cos winds + noise), on the reference's field layout."""
from __future__ import annotations

import numpy as np

from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill

SEED = 20260928


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
