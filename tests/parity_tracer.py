"""Parity of tracer_2d (library + host orchestration) vs the oracle on accumulated fluxes from d_sw."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
import parity_common as P
from gfdl_atmos_cubed_sphere_amd.halo import HaloExchanger
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
from gfdl_atmos_cubed_sphere_amd.lib import Context
from gfdl_atmos_cubed_sphere_amd.tracer2d import tracer_2d
from test_oracle_properties import run_pair


def check_tracer_2d(lib, nx=40, ny=19, npz=4, nq=3, hord=8, q_split=0, trdm=0.0, nord_tr=1, big_courant=False, reverse=False):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    before, after = run_pair(bd, npz, g, True, dt=8.0)    # oracle c_sw + d_sw: realistic mfx, mfy, cx, cy
    rng = np.random.default_rng(17)
    scale = 3.0 if big_courant else 1.0                   # > 1 forces sub-cycling (nsplt > 1)
    if reverse:                                           # the same flow backwards (the default state has u > 0 everywhere)
        scale = -scale
    mfx, mfy = np.asfortranarray(after["mfx"] * scale), np.asfortranarray(after["mfy"] * scale)
    cx, cy = np.asfortranarray(after["cx"] * scale * 3.0), np.asfortranarray(after["cy"] * scale * 3.0)
    dp1 = before["delp"].copy(order="F")
    q = np.asfortranarray(rng.uniform(0, 1, bd.shape("A", npz) + (nq,)))
    ref = dict(q=q.copy(order="F"), dp1=dp1.copy(order="F"), mfx=mfx.copy(order="F"), mfy=mfy.copy(order="F"),
               cx=cx.copy(order="F"), cy=cy.copy(order="F"))
    nsplt_ref = O.tracer_2d(g, npz, nq, ref["q"], ref["dp1"], ref["mfx"], ref["mfy"], ref["cx"], ref["cy"], hord, q_split,
                            nord_tr, trdm)
    ctx = Context(g, npz, lib=lib)
    try:
        halo = HaloExchanger(ctx, 1, 1, 0, 1)
        d = dict(q=ctx.from_host(q), q_nxt=ctx.from_host(np.zeros_like(q)), dp1=ctx.from_host(dp1),
                 dp1_nxt=ctx.from_host(np.zeros_like(dp1)), mfx=ctx.from_host(mfx), mfy=ctx.from_host(mfy),
                 cx=ctx.from_host(cx), cy=ctx.from_host(cy), xfx=ctx.zeros("CX", npz), yfx=ctx.zeros("CY", npz))
        qf, dpf, nsplt = tracer_2d(ctx, halo, d["q"], d["q_nxt"], d["dp1"], d["dp1_nxt"], d["mfx"], d["mfy"], d["cx"],
                                   d["cy"], d["xfx"], d["yfx"], nq, hord, q_split, nord_tr, trdm)
        assert nsplt == nsplt_ref, (nsplt, nsplt_ref)
        tol = 1e-14
        r = (bd.is_, bd.ie, bd.js, bd.je)
        got = qf.download()
        worst = 0.0
        for iq in range(nq):
            worst = max(worst, P.assert_close(f"q{iq}", bd.view(got[:, :, :, iq], "A", *r),
                                              bd.view(ref["q"][:, :, :, iq], "A", *r), tol))
        worst = max(worst, P.assert_close("dp1", bd.view(dpf.download(), "A", *r), bd.view(ref["dp1"], "A", *r), tol))
        for n in ("mfx", "mfy", "cx", "cy"):
            worst = max(worst, P.assert_close(n, d[n].download(), ref[n], tol))
    finally:
        ctx.close()
    return worst, nsplt


def check_fill2d(lib, nx=40, ny=19, npz=4, grid=None, halo_fill=None):
    """fill2D (fv_fill.F90:183-258) on a tracer with negative patches: the two kernels around the halo update of qt against the
    oracle's two halves; halo_fill(qt) fills the halo of a host array (default: the doubly periodic fill)"""
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, False)
    rng = np.random.default_rng(31)
    q = np.asfortranarray(rng.uniform(-0.3, 1.0, bd.shape("A", npz)))
    delp = np.asfortranarray(rng.uniform(500.0, 1500.0, bd.shape("A", npz)))
    if halo_fill is None:
        def halo_fill(a):
            for k in range(a.shape[2]):
                periodic_fill(bd, a[:, :, k], "A")
    ref, qt = q.copy(order="F"), bd.zeros("A", npz)
    O.fill2d_mass(g, npz, ref, delp, qt)
    halo_fill(qt)
    O.fill2d_apply(g, npz, qt, delp, ref)
    ctx = Context(g, npz, lib=lib)
    try:
        d_q, d_dp, d_qt = ctx.from_host(q), ctx.from_host(delp), ctx.zeros("A", npz)
        ctx.fill2d_mass(npz, d_q, d_dp, d_qt)
        h = d_qt.download()
        halo_fill(h)
        d_qt.upload(h)
        ctx.fill2d_apply(npz, d_qt, d_dp, d_q)
        r = (bd.is_, bd.ie, bd.js, bd.je)
        got = d_q.download()
        assert np.any(got != q)
        # the filling moves mass between neighbours: the total of q delp area is what it was (to rounding)
        area = bd.view(g.m["area"], "A", *r)[:, :, None]
        m0, m1 = (bd.view(q, "A", *r) * bd.view(delp, "A", *r) * area).sum(), (bd.view(got, "A", *r) * bd.view(delp, "A", *r) * area).sum()
        assert abs(m1 - m0) <= 1e-12 * abs(m0) or grid is not None
        return P.assert_close("q", bd.view(got, "A", *r), bd.view(ref, "A", *r), 1e-15)
    finally:
        ctx.close()
