"""Parity of the vertical remap (Lagrangian_to_Eulerian) library vs oracle on a deformed-coordinate state."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
import parity_common as P
import parity_nh as N
from fields import smooth_state
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, KAPPA, RDGAS, Context


def remap_state(bd, km, nq, seed=31):
    rng = np.random.default_rng(seed)
    s = N.nh_state(bd, km, seed=seed)
    sig = np.linspace(0.0, 1.0, km + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    delp = np.asfortranarray(s["delp"] * (1.0 + 0.04 * rng.uniform(-1, 1, s["delp"].shape)))
    for k in range(km):
        periodic_fill(bd, delp[:, :, k], "A")
    nx, ny, ng = bd.nx, bd.ny, bd.ng
    pe_full = N.PTOP + np.concatenate([np.zeros(bd.shape("A") + (1,)), np.cumsum(delp, axis=2)], axis=2)
    # pe(is-1:ie+1, km+1, js-1:je+1): k in the middle
    pe = np.asfortranarray(np.transpose(pe_full[ng - 1:ng + nx + 1, ng - 1:ng + ny + 1, :], (0, 2, 1)))
    pc = pe_full[ng:ng + nx, ng:ng + ny, :]
    peln = np.asfortranarray(np.transpose(np.log(pc), (0, 2, 1)))
    pk = np.asfortranarray(np.exp(KAPPA * np.log(pc)))
    w = smooth_state(bd, km, noise=0.2)
    delz = np.asfortranarray(np.diff(s["zh"], axis=2)[ng:ng + nx, ng:ng + ny, :])
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", km) + (nq,)) ** 3) if nq else None
    f = dict(ps=bd.zeros("A"), pe=pe, delp=delp, pkz=bd.zeros("CC", km), pk=pk, u=w["u"], v=w["v"],
             w=np.asfortranarray(w["w"] * 3.0), delz=delz, pt=s["pt"].copy(order="F"), peln=peln,
             omga=np.asfortranarray(rng.uniform(-1, 1, bd.shape("A", km))),
             ws=np.asfortranarray(0.05 * rng.uniform(-1, 1, bd.shape("CC"))))
    if nq:
        f["q"] = q
    return f, ak, bk


MOIST6 = dict(nwat=6, liq_wat=2, rainwat=3, ice_wat=4, snowwat=5, graupel=6, cv_vap=3.0 * 461.50, c_liq=4.218e3,
              c_ice=2.106e3)


def check_remap(lib, nx=20, ny=11, km=12, nq=2, hydrostatic=False, last_step=False, kord_tm=-8, kord=8, adiabatic=True,
                moist_kappa=False, use_cond=False, nwat=6, fill=False, remap_te=False, lds=True):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    moist = moist_kappa or use_cond
    if moist:
        nq = max(nq, 7)
    f, ak, bk = remap_state(bd, km, nq)
    if moist:   # water species: small mixing ratios (q(1) = sphum, 2..6 = condensates)
        f["q"][:, :, :, 0] *= 0.02
        f["q"][:, :, :, 1:6] *= 0.002
        f["q_con"], f["cappa"] = bd.zeros("A", km), bd.zeros("A", km)
    par = dict(last_step=int(last_step), hydrostatic=int(hydrostatic), adiabatic=int(adiabatic), nq=nq, kord_mt=kord,
               kord_wz=kord, kord_tm=kord_tm, sphum=1 if nq else 0, akap=KAPPA, ptop=N.PTOP, rdgas=RDGAS, grav=GRAV,
               cv_air=CP_AIR - RDGAS, r_vir=0.6077, cp=CP_AIR, t_min=184.0, kord_tr=[kord if n % 2 == 0 else 9 for n in range(nq)])
    if fill:   # negative undershoots for fillz to repair: isolated, paired and bottom / top-layer cases
        rng = np.random.default_rng(77)
        qn = f["q"]
        mask = rng.uniform(0, 1, qn.shape) > 0.93
        qn[mask] = -0.3 * qn[mask] - 0.01
        qn[:, :, 0, 0] = -0.02
        qn[:, :, -1, 0] = -0.05
        par["fill"] = 1
    mpar = dict(MOIST6, nwat=nwat, moist_kappa=int(moist_kappa), use_cond=int(use_cond)) if moist else {}
    if moist and nwat == 3:
        mpar.update(liq_wat=2, ice_wat=3, rainwat=0, snowwat=0, graupel=0)
    opar = dict(par, **mpar)
    if remap_te:    # flagstruct%remap_te: hs = phis, te = an A x km work array (fv_mapz.F90:232-286, :348-360, :576-619)
        rng = np.random.default_rng(41)
        f["hs"] = np.asfortranarray(rng.uniform(0.0, 3000.0, bd.shape("A")))
        f["te"] = bd.zeros("A", km)
        opar["remap_te"] = 1
    ref = {k: (v.copy(order="F") if v is not None else None) for k, v in f.items()}
    if hydrostatic:
        ref.pop("w"); ref.pop("delz"); ref.pop("ws")
    O.lagrangian_to_eulerian(g, km, opar, ref, ak, bk)
    if fill:    # fillz really acted, and left no negative values where the column could afford it
        nofill = {k: (v.copy(order="F") if v is not None else None) for k, v in f.items()}
        O.lagrangian_to_eulerian(g, km, dict(opar, fill=0), nofill, ak, bk)
        rr = (bd.is_, bd.ie, bd.js, bd.je)
        assert np.min(bd.view(nofill["q"][:, :, :, 0], "A", *rr)) < 0.0
        assert P.rel_rms(bd.view(nofill["q"][:, :, :, 0], "A", *rr), bd.view(ref["q"][:, :, :, 0], "A", *rr)) > 1e-6
    if moist:   # the moist branches really change the answer
        dry = {k: (v.copy(order="F") if v is not None else None) for k, v in f.items()}
        O.lagrangian_to_eulerian(g, km, par, dry, ak, bk)
        n_chk = "pkz" if moist_kappa else "pt"
        assert P.rel_rms(dry[n_chk], ref[n_chk]) > 1e-6
    import os
    saved = os.environ.pop("FV3_MI355X_REMAP_LDS", None)
    os.environ["FV3_MI355X_REMAP_LDS"] = "2" if lds else "0"      # read when the context is created; 2: the LDS kernels wherever they are built
    try:
        ctx = Context(g, km, lib=lib)
    finally:
        os.environ.pop("FV3_MI355X_REMAP_LDS", None)
        if saved is not None:
            os.environ["FV3_MI355X_REMAP_LDS"] = saved
    try:
        ctx.set_ak_bk(ak, bk)
        # lds: the remap with the column in LDS (csrc/remap_fast.h, the default where it is built and pays; bit-identical to the slab kernels);
        # False: FV3_MI355X_REMAP_LDS=0, the slab kernels (csrc/remap_kernels.h) for every configuration
        d = {k: ctx.from_host(v) for k, v in f.items()}
        if moist:
            ctx.set_moist(mpar, d["q_con"], d["cappa"])
        if remap_te:
            ctx.set_remap_te(True, d["hs"], d["te"])
        ctx.lagrangian_to_eulerian(par, d["ps"], d["pe"], d["delp"], d["pkz"], d["pk"], d["u"], d["v"],
                                   None if hydrostatic else d["w"], None if hydrostatic else d["delz"], d["pt"],
                                   d.get("q"), d["peln"], d["omga"], None if hydrostatic else d["ws"])
        tol = 1e-14
        r = (bd.is_, bd.ie, bd.js, bd.je)
        names = [("pt", "A", r), ("delp", "A", r), ("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)),
                 ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)), ("ps", "A", r)]
        if not hydrostatic:
            names.append(("w", "A", r))
        if last_step:
            names.append(("omga", "A", r))
        worst = 0.0
        for n, kind, rr in names:
            worst = max(worst, P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), tol))
        for n in ("pkz", "pk", "peln") + (() if hydrostatic else ("delz",)):
            worst = max(worst, P.assert_close(n, d[n].download(), ref[n], tol))
        worst = max(worst, P.assert_close("pe", d["pe"].download()[1:-1, :, 1:-1], ref["pe"][1:-1, :, 1:-1], tol))
        if moist_kappa:
            for n in ("q_con", "cappa"):
                worst = max(worst, P.assert_close(n, bd.view(d[n].download(), "A", *r), bd.view(ref[n], "A", *r), tol))
        if remap_te:   # the remapped energy itself, and (the remap moved something) not the energy before the remap
            worst = max(worst, P.assert_close("te", bd.view(d["te"].download(), "A", *r), bd.view(ref["te"], "A", *r), tol))
            plain = {k: (v.copy(order="F") if v is not None else None) for k, v in f.items()}
            if hydrostatic:
                plain.pop("w"); plain.pop("delz"); plain.pop("ws")
            O.lagrangian_to_eulerian(g, km, dict(opar, remap_te=0, kord_tm=(kord_tm if kord_tm else -9)), plain, ak, bk)
            e = P.rel_rms(bd.view(plain["pt"], "A", *r), bd.view(ref["pt"], "A", *r))
            assert 1e-7 < e < 2e-2, e     # another scheme for the same quantity: close, not equal
        if nq:
            got = d["q"].download()
            for iq in range(nq):
                worst = max(worst, P.assert_close(f"q{iq}", bd.view(got[:, :, :, iq], "A", *r),
                                                  bd.view(ref["q"][:, :, :, iq], "A", *r), tol))
        # conservation identity of the remap (fv_operators.F90:130): column mass of each tracer
        if nq:
            dp_old = bd.view(f["delp"], "A", *r)
            dp_new = bd.view(d["delp"].download(), "A", *r)
            for iq in range(nq):
                m0 = np.sum(bd.view(f["q"][:, :, :, iq], "A", *r) * dp_old, axis=2)
                m1 = np.sum(bd.view(got[:, :, :, iq], "A", *r) * dp_new, axis=2)
                assert np.max(np.abs(m1 - m0) / np.abs(m0)) < 1e-12
    finally:
        ctx.close()
    return worst
