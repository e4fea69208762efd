"""Cubed-sphere parity: the library's grid_type < 3 kernels against the oracle, face by face, on the global smooth state of
cubed_common (halos filled by the emulated six-face updates)."""
from __future__ import annotations

import numpy as np

import cubed_common as CC
import oracle_lib as O
import parity_common as P
from gfdl_atmos_cubed_sphere_amd import lib as L
from gfdl_atmos_cubed_sphere_amd.lib import Context
from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT


def check_c_sw(lib, npx=13, npz=3, hydrostatic=False, faces=range(6), dt2=300.0, nord=1):
    cs, gs, st = CC.global_state(npx, npz, hydrostatic)
    worst = 0.0
    for t in faces:
        g, bd = gs[t], gs[t].bd
        f = {k: v.copy(order="F") for k, v in st[t].items()}
        for n, kind in CSW_OUT:
            f[n] = bd.zeros(kind, npz)
        O.c_sw_3d(g, npz, f, nord=nord, dt2=dt2, hydrostatic=hydrostatic)
        ctx = Context(g, npz, lib=lib)
        try:
            d = {k: ctx.from_host(v) for k, v in st[t].items()}
            for n, kind in CSW_OUT:
                d[n] = ctx.zeros(kind, npz)
            ctx.c_sw(d["delpc"], d["delp"], d["ptc"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"],
                     d.get("wc") if not hydrostatic else None, d["ut"], d["vt"], d["divg_d"], nord, dt2, hydrostatic)
            rng = P.csw_valid_ranges(bd)
            rng["divg_d"] = (bd.is_, bd.ie + 1, bd.js, bd.je + 1)      # the cubed-sphere form fills the compute corners only
            for n, kind in CSW_OUT:
                if hydrostatic and n == "wc":
                    continue
                if n == "divg_d" and nord == 0:
                    continue
                got, ref = bd.view(d[n].download(), kind, *rng[n]), bd.view(f[n], kind, *rng[n])
                worst = max(worst, P.assert_close(f"face {t + 1} {n}", got, ref))
        finally:
            ctx.close()
    return worst


def check_fv_tp_2d(lib, hord, npx=13, nk=3, faces=range(6), mass_flux=False, seed=4, nord=-1, damp_c=0.0):
    """fv_tp_2d on every face: q and the Courant numbers / area fluxes of a real c_sw -> d_sw step of the oracle"""
    cs, gs, before, after = CC.oracle_pair(npx, nk, dt=600.0, hydrostatic=True)
    rng = np.random.default_rng(seed)
    worst = 0.0
    for t in faces:
        g, bd = gs[t], gs[t].bd
        q = before[t]["pt"].copy(order="F")
        q[..., 0] = np.asfortranarray(rng.uniform(0.0, 1.0, q.shape[:2]) ** 3)      # a rough field on one level
        CCq = q
        a = after[t]
        mfx = np.asfortranarray(rng.uniform(-1, 1, bd.shape("FX", nk)) * 1e5) if mass_flux else None
        mfy = np.asfortranarray(rng.uniform(-1, 1, bd.shape("FY", nk)) * 1e5) if mass_flux else None
        mass = before[t]["delp"].copy(order="F") if (mass_flux and nord >= 0) else None      # deln_flux weighted with delp
        ra_x, ra_y = bd.zeros("RX", nk), bd.zeros("RY", nk)
        ng, nx = bd.ng, bd.nx
        ra_x[...] = g.m["area"][ng:ng + nx, :, None] + a["xfx"][:-1, :, :] - a["xfx"][1:, :, :]
        ra_y[...] = g.m["area"][:, ng:ng + nx, None] + a["yfx"][:, :-1, :] - a["yfx"][:, 1:, :]
        fx_ref, fy_ref = bd.zeros("FX", nk), bd.zeros("FY", nk)
        for k in range(nk):
            sl = lambda x: None if x is None else np.asfortranarray(x[:, :, k])      # noqa: E731
            qk = np.asfortranarray(CCq[:, :, k]).copy(order="F")
            fx, fy = O.fv_tp_2d(g, qk, sl(a["crx"]), sl(a["cry"]), hord, sl(a["xfx"]), sl(a["yfx"]), sl(ra_x), sl(ra_y),
                                mfx=sl(mfx), mfy=sl(mfy), mass=sl(mass), nord=nord, damp_c=damp_c)
            fx_ref[:, :, k], fy_ref[:, :, k] = fx, fy
        ctx = Context(g, nk, lib=lib)
        try:
            dfx, dfy = ctx.zeros("FX", nk), ctx.zeros("FY", nk)
            ctx.fv_tp_2d(ctx.from_host(CCq), ctx.from_host(a["crx"]), ctx.from_host(a["cry"]), hord, dfx, dfy, ctx.from_host(a["xfx"]),
                         ctx.from_host(a["yfx"]), ctx.from_host(ra_x), ctx.from_host(ra_y),
                         None if mfx is None else ctx.from_host(mfx), None if mfy is None else ctx.from_host(mfy),
                         None if mass is None else ctx.from_host(mass), nord, damp_c)
            worst = max(worst, P.assert_close(f"face {t + 1} fx", dfx.download(), fx_ref))
            worst = max(worst, P.assert_close(f"face {t + 1} fy", dfy.download(), fy_ref))
        finally:
            ctx.close()
    return worst


def check_d_sw(lib, npx=13, npz=3, hydrostatic=True, faces=range(6), dt=600.0, par_over=None, flags=None, use_cond=False,
               grid_flags=None):
    """c_sw (oracle) on every face -> emulated halo updates of uc, vc, divg_d -> d_sw by the oracle and by the library.
    grid_flags: do_diss_est / prevent_diss_cooling of the gridstruct (the sphere's grids are shared: set, run, restore)"""
    if grid_flags:
        _, gs_ = CC.sphere(npx)
        old = [{k: getattr(g, k) for k in grid_flags} for g in gs_]
        for g in gs_:
            for k, v in grid_flags.items():
                setattr(g, k, v)
        try:
            return check_d_sw(lib, npx, npz, hydrostatic, faces, dt, par_over, flags, use_cond)
        finally:
            for g, o in zip(gs_, old):
                for k, v in o.items():
                    setattr(g, k, v)
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
    cs, gs, before, after = CC.oracle_pair(npx, npz, dt=dt, hydrostatic=hydrostatic, par_over=par_over, flags=flags, use_cond=use_cond)
    fl = DynFlags(**(flags or {}))
    lev = level_coefficients(npz, fl)
    from gfdl_atmos_cubed_sphere_amd.synthetic import DSW_PAR
    par = dict(DSW_PAR)
    par.update(dt=dt, hydrostatic=int(hydrostatic), use_cond=int(use_cond))
    par.update({k: v for k, v in (par_over or {}).items() if k in par})
    worst = {}
    for t in faces:
        g, bd = gs[t], gs[t].bd
        b, a = before[t], after[t]
        ctx = Context(g, npz, lib=lib)
        try:
            ctx.dsw_levels(lev)
            d = {k: ctx.from_host(v) for k, v in b.items() if k not in ("wc",)}
            for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"),
                            ("yfx", "CY")):
                d[n] = ctx.zeros(kind, npz)
            out = {n: ctx.zeros(kind, npz) for n, kind in (("delp_out", "A"), ("pt_out", "A"), ("u_out", "U"), ("v_out", "V"),
                                                           ("w_out", "A"), ("heat_s", "CC"), ("diss_e", "CC"), ("delpc_o", "A"), ("qc_out", "A"))}
            ctx.d_sw(par, out["delpc_o"], d["delp"], d["pt"], d["u"], d["v"], d.get("w"), d["uc"], d["vc"], d["ua"], d["va"],
                     d["divg_d"], d["mfx"], d["mfy"], d["cx"], d["cy"], d["crx"], d["cry"], d["xfx"], d["yfx"], d.get("q_con"),
                     out["delp_out"], out["pt_out"], out["u_out"], out["v_out"], None if hydrostatic else out["w_out"],
                     out["qc_out"] if use_cond else None, out["heat_s"], out["diss_e"])
            i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
            cmp = [("crx", d["crx"], "CX", None), ("cry", d["cry"], "CY", None), ("xfx", d["xfx"], "CX", None),
                   ("yfx", d["yfx"], "CY", None), ("cx", d["cx"], "CX", None), ("cy", d["cy"], "CY", None),
                   ("mfx", d["mfx"], "FX", None), ("mfy", d["mfy"], "FY", None),
                   ("delp", out["delp_out"], "A", (i0, i1, j0, j1)), ("pt", out["pt_out"], "A", (i0, i1, j0, j1)),
                   ("u", out["u_out"], "U", (i0, i1, j0, j1 + 1)), ("v", out["v_out"], "V", (i0, i1 + 1, j0, j1))]
            if not hydrostatic:
                cmp.append(("w", out["w_out"], "A", (i0, i1, j0, j1)))
            cmp.append(("heat_source", out["heat_s"], "CC", None))      # w damping of the sponge levels, d_con heating
            if g.do_diss_est:
                cmp.append(("diss_est", out["diss_e"], "CC", None))
            if use_cond:
                cmp.append(("q_con", out["qc_out"], "A", (i0, i1, j0, j1)))
            for name, dev, kind, r in cmp:
                got, ref = dev.download(), a[name]
                if r is not None:
                    got, ref = bd.view(got, kind, *r), bd.view(ref, kind, *r)
                worst[name] = max(worst.get(name, 0.0), P.assert_close(f"face {t + 1} {name}", got, ref))
        finally:
            ctx.close()
    return worst


def check_substeps_hydrostatic(lib, npx=13, npz=4, n_split=2, bdt=600.0, flags=None):
    """the hydrostatic acoustic substep loop on the whole sphere: six contexts behind dyn_core.DynCore (cubed_dyn.MultiContext,
    halo updates by device gathers) against the six-face orchestration of the oracle"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    cs, gs, st = CC.hydro_state(npx, npz)
    fl = DynFlags(n_split=n_split, hydrostatic=True, **dict(dict(d_ext=0.0), **(flags or {})))
    ref = CC.oracle_substeps_hydro(cs, gs, fl, st, bdt, npz)
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = {}
    try:
        sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
        dp0 = np.diff(fl.ptop + (1.0e5 - fl.ptop) * sig)
        dc = DynCore(mctx, fl, dp0, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
        z = [np.zeros_like(s["delp"]) for s in st]
        bd = gs[0].bd
        dz = [bd.zeros("CC", npz) for _ in st]
        dc.set_state([s["u"] for s in st], [s["v"] for s in st], z, [s["delp"] for s in st], [s["pt"] for s in st], dz,
                     [s["phis"] for s in st])
        dc.run(bdt)
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r)):
            got = dc.d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", bd.view(got[t], kind, *rr), bd.view(ref[t][n], kind, *rr), 1e-13))
        for n in ("mfx", "mfy", "cx", "cy", "pk", "pkz", "peln"):
            got = dc.d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", got[t], ref[t][n], 1e-13))
    finally:
        mctx.close()
    return worst


def check_substeps_nh(lib, npx=13, npz=5, n_split=2, bdt=300.0, flags=None, tol=1e-12):
    """the nonhydrostatic acoustic substep loop on the whole sphere: six contexts behind dyn_core.DynCore against the six-face
    orchestration of the oracle (c_sw, update_dz_c, riem_solver_c, p_grad_c, d_sw, update_dz_d, riem_solver3, nh_p_grad)"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    cs, gs, st = CC.nh_state(npx, npz)
    fl = DynFlags(n_split=n_split, hydrostatic=False, **(flags or {}))
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    dp0 = np.diff(fl.ptop + (1.0e5 - fl.ptop) * sig)
    ref = CC.oracle_substeps_nh(cs, gs, fl, dp0, st, bdt, npz)
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = {}
    try:
        dc = DynCore(mctx, fl, dp0, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
        dc.set_state([s["u"] for s in st], [s["v"] for s in st], [s["w"] for s in st], [s["delp"] for s in st],
                     [s["pt"] for s in st], [s["delz"] for s in st], [s["phis"] for s in st])
        dc.run(bdt)
        bd = gs[0].bd
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r), ("w", "A", r), ("zh", "A", r)):
            got = dc.d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", bd.view(got[t], kind, *rr), bd.view(ref[t][n], kind, *rr), tol))
        for n in ("delz", "mfx", "mfy", "cx", "cy", "pk", "peln"):
            got = dc.d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", got[t], ref[t][n], tol))
    finally:
        mctx.close()
    return worst


def tracer_fields(cs, npz, nq):
    """nq smooth positive tracer fields per face as functions of position (A x npz x nq, halo included)"""
    out = []
    for t in range(6):
        a3 = cs.grids[t]["agrid3"]
        lev = (np.arange(npz) / max(npz - 1, 1))[None, None, :]
        out.append(np.asfortranarray(np.stack(
            [(1.0 + iq) * (1.0 + 0.3 * np.sin(2.0 * a3[..., 0:1] * (1 + iq % 3) + 4.0 * lev) * np.cos(3.0 * a3[..., 1:2] - iq)
                           + 0.2 * a3[..., 2:3] ** 2 * lev) for iq in range(nq)], axis=-1)))
    return out


def check_jw_step(lib, npx=13, npz=79, k_split=1, n_split=2, bdt=600.0, tol=1e-12, hydrostatic=True, nq=0, face_streams=False, flags=None,
                  native_halo=False, fill2d=(), q_shift=0.0, remap_te=False, kord_tm=-8):
    """BASELINE configs[1] in small: the Jablonowski-Williamson baroclinic wave (test_case = 13) on the whole cubed sphere,
    hydrostatic, the reference's L79 levels (set_eta), one dt_atmos = k_split x (n_split substeps + vertical remap) on six
    device contexts against the six-face orchestration of the oracle"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.test_cases import set_eta
    cs, gs = CC.sphere(npx)
    if npz in (79, 127):
        ak, bk, ks, ptop = set_eta(npz)
    else:
        sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
        ptop = 300.0
        ak, bk = ptop * (1.0 - sig), sig.copy()
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=hydrostatic, rdgas=L.RDGAS, grav=L.GRAV)
    CC.exchange(cs, st, ("phis",), "A")          # the model gets phis with its halo filled (init_case: mpp_update_domains(phis))
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, d_ext=0.0, ptop=float(ak[0]), **(flags or {}))
    # T -> theta: pt = T / pkz with the hydrostatic pkz of the initial state (fv_dynamics.F90:323-329, the host's job here)
    bd = gs[0].bd
    ng, nx = bd.ng, bd.nx
    for s in st:
        pe = ak[0] + np.concatenate([np.zeros(s["delp"].shape[:2] + (1,)), np.cumsum(s["delp"], axis=2)], axis=2)
        c = (slice(ng, ng + nx), slice(ng, ng + nx))
        pk = O.fexp(fl.akap * O.flog(pe[c])).reshape(pe[c].shape)
        peln = O.flog(pe[c]).reshape(pe[c].shape)
        pkz = (pk[:, :, 1:] - pk[:, :, :-1]) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
        if not hydrostatic:          # fv_dynamics.F90:385-394: pkz = (rdg * delp * pt / delz) ** kappa
            arg = (-fl.rdgas / fl.grav) * s["delp"][c] * s["pt"][c] / s["delz"]
            pkz = O.fexp(fl.akap * O.flog(arg)).reshape(arg.shape)
        s["pt"][c] = s["pt"][c] / pkz
    if face_streams:      # every face on a HIP stream of its own; the halo gathers join / fork them
        import torch
        streams = [torch.cuda.Stream() for _ in gs] if face_streams != "one" else [torch.cuda.Stream()] * 6
        mctx = MultiContext([Context(g, npz, lib=lib, stream=st_.cuda_stream) for g, st_ in zip(gs, streams)])
    else:
        mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = {}
    try:
        if native_halo:      # the cube-edge exchange behind the C ABI: every message through fv3_cube_halo_start / _complete
            from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeNativeAdapter
            halo = CubeNativeAdapter(mctx, range(6), [0] * 6)
        else:
            halo = CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx))
        fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=k_split, halo=halo, fill2d=fill2d, moist_phys=bool(fill2d), remap_te=remap_te,
                        kord_tm=kord_tm)
        rpar = dict(fv.remap_par, remap_te=int(remap_te))
        q0 = tracer_fields(cs, npz, nq) if nq else None
        if q_shift:          # patches of negative tracer mass for fill2D
            q0 = [np.asfortranarray(x - q_shift) for x in q0]
        if hydrostatic:
            ref = CC.oracle_fv_step_hydro(cs, gs, fl, st, ak, bk, bdt, k_split, rpar, npz, q=q0, fill2d=fill2d)
            z = [np.zeros_like(s["delp"]) for s in st]
            dz = [bd.zeros("CC", npz) for _ in st]
        else:
            dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
            ref = CC.oracle_fv_step_nh(cs, gs, fl, dp_ref, st, ak, bk, bdt, k_split, rpar, npz, q=q0, fill2d=fill2d)
            z, dz = [s["w"] for s in st], [s["delz"] for s in st]
        fv.dc.set_state([s["u"] for s in st], [s["v"] for s in st], z, [s["delp"] for s in st], [s["pt"] for s in st], dz,
                        [s["phis"] for s in st])
        if nq:
            fv.set_tracers(q0)
        fv.step(bdt)
        d = fv.dc.d
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r), ("ps", "A", r)):
            got = d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", bd.view(got[t], kind, *rr), bd.view(ref[t][n], kind, *rr), tol))
        for n in ("pkz", "pk", "peln") + (() if hydrostatic else ("delz",)):
            got = d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", got[t], ref[t][n], tol))
        if not hydrostatic:
            got = d["w"].download()
            for t in range(6):
                worst["w"] = max(worst.get("w", 0.0), P.assert_close(f"face {t + 1} w", bd.view(got[t], "A", *r), bd.view(ref[t]["w"], "A", *r), tol))
        if nq:
            got = d["q"].download()
            for t in range(6):
                for iq in range(nq):
                    worst["q"] = max(worst.get("q", 0.0), P.assert_close(f"face {t + 1} q{iq}", bd.view(got[t][:, :, :, iq], "A", *r),
                                                                           bd.view(ref[t]["q"][:, :, :, iq], "A", *r), tol))
        # cubed_to_latlon of the new winds (fv_dynamics.F90:911): halo update of u, v (c2l_ord4), then the rotation to (east, north)
        fv.cubed_to_latlon()
        ru, rv = [x["u"].copy(order="F") for x in ref], [x["v"].copy(order="F") for x in ref]
        cs.topo.update("D", (ru, rv))
        gua, gva = d["ua"].download(), d["va"].download()
        for t in range(6):
            ua, va = bd.zeros("A", npz), bd.zeros("A", npz)
            O.c2l(gs[t], npz, fv.c2l_ord, ru[t], rv[t], ua, va)
            worst["ua"] = max(worst.get("ua", 0.0), P.assert_close(f"face {t + 1} ua", bd.view(gua[t], "A", *r), bd.view(ua, "A", *r), tol))
            worst["va"] = max(worst.get("va", 0.0), P.assert_close(f"face {t + 1} va", bd.view(gva[t], "A", *r), bd.view(va, "A", *r), 10 * tol))
        dp = d["delp"].download()
        s = slice(ng, ng + nx)
        worst["finite"] = float(all(np.isfinite(x[s, s, :]).all() for x in dp))
    finally:
        mctx.close()
    return worst


def check_sphere_properties(lib, npx=97, npz=127, hydrostatic=False, k_split=1, n_split=2, bdt=225.0, nq=0, flags=None):
    """size-independent checks of a whole-sphere step at sizes the oracle cannot reach: everything finite, the global air mass
    sum(area * delp) kept to rounding (flux form; both faces of a cube edge compute the same edge flux), the winds on the shared
    cube edges equal on both faces (mpp_get_boundary), the state moved"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.test_cases import set_eta
    cs, gs = CC.sphere(npx)
    ak, bk, ks, ptop = set_eta(npz)
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=hydrostatic, rdgas=L.RDGAS, grav=L.GRAV)
    CC.exchange(cs, st, ("phis",), "A")          # the model gets phis with its halo filled (init_case: mpp_update_domains(phis))
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, d_ext=0.0, ptop=float(ak[0]), **(flags or {}))
    bd = gs[0].bd
    ng, nx = bd.ng, bd.nx
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    for s_ in st:
        if hydrostatic:
            pe = ak[0] + np.concatenate([np.zeros(s_["delp"].shape[:2] + (1,)), np.cumsum(s_["delp"], axis=2)], axis=2)[c]
            peln = np.log(pe)
            pkz = (pe[:, :, 1:] ** fl.akap - pe[:, :, :-1] ** fl.akap) / (fl.akap * (peln[:, :, 1:] - peln[:, :, :-1]))
        else:
            pkz = ((-fl.rdgas / fl.grav) * s_["delp"][c] * s_["pt"][c] / s_["delz"]) ** fl.akap
        s_["pt"][c] = s_["pt"][c] / pkz
    area = [g.m["area"][c] for g in gs]
    mass0 = sum(float(np.sum(a[:, :, None] * s_["delp"][c])) for a, s_ in zip(area, st))
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    out = {}
    try:
        fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=k_split, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
        if nq:
            q0 = tracer_fields(cs, npz, nq)
            fv.set_tracers(q0)
            qm0 = [sum(float(np.sum(a[:, :, None] * s_["delp"][c] * x[c][..., iq])) for a, s_, x in zip(area, st, q0)) for iq in range(nq)]
            del q0
        z = [np.zeros_like(s_["delp"]) for s_ in st] if hydrostatic else [s_["w"] for s_ in st]
        dz = [bd.zeros("CC", npz) for _ in st] if hydrostatic else [s_["delz"] for s_ in st]
        fv.dc.set_state([s_["u"] for s_ in st], [s_["v"] for s_ in st], z, [s_["delp"] for s_ in st], [s_["pt"] for s_ in st], dz,
                        [s_["phis"] for s_ in st])
        fv.step(bdt)
        d = fv.dc.d
        dp = d["delp"].download()
        u, v = d["u"].download(), d["v"].download()
        out["finite"] = float(all(np.isfinite(x[c]).all() for x in dp) and all(np.isfinite(bd.view(x, "U", bd.is_, bd.ie, bd.js, bd.je + 1)).all() for x in u))
        mass1 = sum(float(np.sum(a[:, :, None] * x[c])) for a, x in zip(area, dp))
        out["mass_drift"] = abs(mass1 - mass0) / mass0
        out["moved"] = max(float(np.max(np.abs(x[c] - s_["delp"][c]))) for x, s_ in zip(dp, st))
        if nq:      # the tracer mass sum(area * delp * q) of every tracer is kept by the flux form + the conservative remap
            q1 = d["q"].download()
            qm1 = [sum(float(np.sum(a[:, :, None] * x[c] * y[c][..., iq])) for a, x, y in zip(area, dp, q1)) for iq in range(nq)]
            out["tracer_mass_drift"] = max(abs(b - a) / abs(a) for a, b in zip(qm0, qm1))
            out["tracer_finite"] = float(all(np.isfinite(y[c]).all() for y in q1))
        # the shared edges: apply the table update for the boundary points to copies and compare
        uu, vv = [x.copy(order="F") for x in u], [x.copy(order="F") for x in v]
        cs.topo.update("Dedge", (uu, vv))
        out["edge_mismatch"] = max(max(float(np.max(np.abs(bd.view(a, "U", bd.is_, bd.ie, bd.js, bd.je + 1) - bd.view(b, "U", bd.is_, bd.ie, bd.js, bd.je + 1))))
                                       for a, b in zip(u, uu)),
                                   max(float(np.max(np.abs(bd.view(a, "V", bd.is_, bd.ie + 1, bd.js, bd.je) - bd.view(b, "V", bd.is_, bd.ie + 1, bd.js, bd.je))))
                                       for a, b in zip(v, vv)))
    finally:
        mctx.close()
    return out


def check_tracer_2d(lib, npx=13, npz=4, nq=3, hord=8, q_split=0, courant_scale=1.0, dt=600.0, nord_tr=0, trdm=0.0):
    """tracer_2d on the whole sphere with the mass fluxes / Courant numbers of one c_sw -> d_sw step of the oracle; a
    courant_scale > 1 makes the levels sub-cycle (nsplt > 1, different ksplt per level)"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.tracer2d import tracer_2d
    cs, gs, before, after = CC.oracle_pair(npx, npz, dt=dt, hydrostatic=True)
    bd = gs[0].bd
    lev_scale = courant_scale * (1.0 + 0.5 * np.arange(npz) / npz) if courant_scale != 1.0 else np.ones(npz)
    inp = []
    for t in range(6):
        a3 = cs.grids[t]["agrid3"]
        q = np.stack([np.stack([CC.scalar(a3, k + 3 * iq, npz, 1.0 + iq, 0.3) * (1.0 + 0.05 * CC._ripple(a3 * (1.0 + 0.1 * iq))) for k in range(npz)], axis=-1)
                      for iq in range(nq)], axis=-1)
        x = dict(q=np.asfortranarray(q), dp1=before[t]["delp"].copy(order="F"))
        for n in ("mfx", "mfy", "cx", "cy"):
            x[n] = np.asfortranarray(after[t][n] * lev_scale[None, None, :])
        inp.append(x)
    ref = [{k: v.copy(order="F") for k, v in x.items()} for x in inp]
    nsplt = CC.oracle_tracer_2d(cs, gs, npz, nq, [r["q"] for r in ref], [r["dp1"] for r in ref], [r["mfx"] for r in ref],
                                [r["mfy"] for r in ref], [r["cx"] for r in ref], [r["cy"] for r in ref], hord, q_split, nord_tr, trdm)
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = {"nsplt": float(nsplt)}
    try:
        halo = CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx))
        d = {n: mctx.from_host([x[n] for x in inp]) for n in ("q", "dp1", "mfx", "mfy", "cx", "cy")}
        d["q_nxt"], d["dp1_nxt"] = mctx.from_host([x["q"] * 0 for x in inp]), mctx.zeros("A", npz)
        d["xfx"], d["yfx"] = mctx.zeros("CX", npz), mctx.zeros("CY", npz)
        q, dp1, ns = tracer_2d(mctx, halo, d["q"], d["q_nxt"], d["dp1"], d["dp1_nxt"], d["mfx"], d["mfy"], d["cx"], d["cy"],
                               d["xfx"], d["yfx"], nq, hord, q_split, nord_tr, trdm)
        assert ns == nsplt, (ns, nsplt)
        got = q.download()
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for t in range(6):
            for iq in range(nq):
                worst["q"] = max(worst.get("q", 0.0), P.assert_close(f"face {t + 1} q{iq}", bd.view(got[t][:, :, :, iq], "A", *r),
                                                                       bd.view(ref[t]["q"][:, :, :, iq], "A", *r)))
        # mass-weighted tracer content is conserved by the flux form
    finally:
        mctx.close()
    return worst


def check_del2_cubed(lib, npx=13, npz=3, nmax=3, faces=range(6)):
    """del2_cubed (dyn_core.F90:2356) on the faces: the mean over the three cells around each cube corner, copy_corners before the
    differences, three smoothing passes on shrinking boxes"""
    cs, gs = CC.sphere(npx)
    q = []
    for t in range(6):
        a3 = cs.grids[t]["agrid3"]
        q.append(np.asfortranarray(np.stack([CC.scalar(a3, k, npz, 1.0, 0.5) * (1.0 + 0.3 * CC._ripple(a3)) for k in range(npz)], axis=-1)))
    cs.topo.update("A", q)
    worst = 0.0
    for t in faces:
        g, bd = gs[t], gs[t].bd
        ref = q[t].copy(order="F")
        O.del2_cubed(g, npz, 0.20 * g.da_min, nmax, ref)
        ctx = Context(g, npz, lib=lib)
        try:
            d = ctx.from_host(q[t])
            ctx.del2_cubed(d, 0.20 * g.da_min, nmax)
            r = (bd.is_, bd.ie, bd.js, bd.je)
            worst = max(worst, P.assert_close(f"face {t + 1} del2_cubed", bd.view(d.download(), "A", *r), bd.view(ref, "A", *r)))
            assert np.max(np.abs(bd.view(ref, "A", *r) - bd.view(q[t], "A", *r))) > 1e-6
        finally:
            ctx.close()
    return worst


def check_c2l(lib, ord_, npx=13, npz=3, faces=range(6)):
    """cubed_to_latlon (c2l_ord2 / c2l_ord4, fv_grid_utils.F90:2330-2560) on the faces: D-grid winds -> (east, north) at the cell
    centres; for the solid-body part of the test wind the result is the analytic zonal / meridional wind to discretisation error"""
    cs, gs, st = CC.global_state(npx, npz, hydrostatic=True)
    worst = 0.0
    for t in faces:
        g, bd = gs[t], gs[t].bd
        ua, va = bd.zeros("A", npz), bd.zeros("A", npz)
        O.c2l(g, npz, ord_, st[t]["u"], st[t]["v"], ua, va)
        ctx = Context(g, npz, lib=lib)
        try:
            d_ua, d_va = ctx.zeros("A", npz), ctx.zeros("A", npz)
            ctx.c2l(ord_, ctx.from_host(st[t]["u"]), ctx.from_host(st[t]["v"]), d_ua, d_va)
            r = (bd.is_, bd.ie, bd.js, bd.je)
            worst = max(worst, P.assert_close(f"face {t + 1} ua", bd.view(d_ua.download(), "A", *r), bd.view(ua, "A", *r)))
            worst = max(worst, P.assert_close(f"face {t + 1} va", bd.view(d_va.download(), "A", *r), bd.view(va, "A", *r)))
            # physics: the winds are those of CC.wind at the centres
            a3 = cs.grids[t]["agrid3"]
            lon, lat = cs.grids[t]["agrid"][..., 0], cs.grids[t]["agrid"][..., 1]
            e = np.stack([-np.sin(lon), np.cos(lon), np.zeros_like(lon)], -1)
            n = np.stack([-np.sin(lat) * np.cos(lon), -np.sin(lat) * np.sin(lon), np.cos(lat)], -1)
            w3 = CC.wind(a3)
            c = (slice(bd.ng, bd.ng + bd.nx), slice(bd.ng, bd.ng + bd.nx))
            ue, vn = np.sum(w3 * e, -1)[c], np.sum(w3 * n, -1)[c]
            assert np.max(np.abs(bd.view(ua, "A", *r)[:, :, 0] - ue)) < 0.12 * np.max(np.abs(ue)), t      # C12: coarse + the 2 % ripple
            assert np.max(np.abs(bd.view(va, "A", *r)[:, :, 0] - vn)) < 0.12 * max(np.max(np.abs(vn)), np.max(np.abs(ue))), t
        finally:
            ctx.close()
    return worst


def check_rayleigh(lib, npx=13, npz=8, hydrostatic=False, conserve=True, tau=0.02, rf_cutoff=30.e2):
    """Rayleigh_Friction on the whole sphere (fv_dynamics.F90:1126-1264): u2f from the cubed-sphere cubed_to_latlon on every face,
    its halo update across the cube edges, the frictional heating and the implicit damping of u, v, w -- oracle vs library"""
    from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, RDGAS
    cs, gs, st = CC.global_state(npx, npz, hydrostatic=hydrostatic)
    ptop = 100.0
    pm = ptop * np.exp(np.linspace(0.1, 5.0, npz))
    rf, kmax = O.rayleigh_rf(npz, 225.0, tau, rf_cutoff, ptop, pm)
    assert 0 < kmax < npz
    rng = np.random.default_rng(31)
    bd = gs[0].bd
    ref = []
    for t in range(6):
        f = {k: st[t][k].copy(order="F") for k in ("u", "v", "pt")}
        f["w"] = None if hydrostatic else np.asfortranarray(st[t]["w"] * 10.0)
        f["delz"] = None if hydrostatic else np.asfortranarray(-rng.uniform(200., 400., bd.shape("CC", npz)))
        f["ua"], f["va"], f["u2f"] = bd.zeros("A", npz), bd.zeros("A", npz), bd.zeros("A", kmax)
        ref.append(f)
    start = [{k: (None if v is None else v.copy(order="F")) for k, v in f.items()} for f in ref]
    for t in range(6):
        f = ref[t]
        O.rayleigh_u2f(gs[t], kmax, hydrostatic, f["u"], f["v"], f["w"], f["ua"], f["va"], f["u2f"])
    CC.exchange(cs, ref, ("u2f",), "A")                         # mpp_update_domains(u2f), :1208
    for t in range(6):
        f = ref[t]
        O.rayleigh_apply(gs[t], kmax, conserve, hydrostatic, CP_AIR, RDGAS, ptop, pm, rf, f["u2f"], f["pt"], f["delz"], f["u"], f["v"], f["w"])
    assert np.max(np.abs(ref[0]["u"][:, :, 0] - start[0]["u"][:, :, 0])) > 1e-4 * np.max(np.abs(start[0]["u"][:, :, 0]))
    ctxs = [Context(g, npz, lib=lib) for g in gs]
    worst = 0.0
    try:
        dev = []
        for t in range(6):
            s, c = start[t], ctxs[t]
            d = {k: (None if s[k] is None else c.from_host(s[k])) for k in ("u", "v", "pt", "w", "delz")}
            d["ua"], d["va"], d["u2f"] = c.zeros("A", npz), c.zeros("A", npz), c.zeros("A", npz)
            c.rayleigh_u2f(kmax, hydrostatic, d["u"], d["v"], d["w"], d["ua"], d["va"], d["u2f"])
            dev.append(d)
        host = [{"u2f": d["u2f"].download()} for d in dev]
        CC.exchange(cs, host, ("u2f",), "A")
        for t in range(6):
            d, c = dev[t], ctxs[t]
            d["u2f"].upload(host[t]["u2f"])
            c.rayleigh_apply(kmax, conserve, hydrostatic, CP_AIR, RDGAS, ptop, pm[:kmax], rf[:kmax], d["u2f"], d["pt"], d["delz"], d["u"],
                             d["v"], d["w"])
            i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
            for n, kind, r in (("u", "U", (i0, i1, j0, j1 + 1)), ("v", "V", (i0, i1 + 1, j0, j1)), ("pt", "A", (i0, i1, j0, j1))) + \
                    (() if hydrostatic else (("w", "A", (i0, i1, j0, j1)),)):
                worst = max(worst, P.assert_close(f"face {t + 1} {n}", bd.view(d[n].download(), kind, *r), bd.view(ref[t][n], kind, *r), 1e-14))
            if not hydrostatic:
                worst = max(worst, P.assert_close(f"face {t + 1} delz", d["delz"].download(), ref[t]["delz"], 1e-14))
    finally:
        for c in ctxs:
            c.close()
    return worst


def check_jw_step_moist(lib, npx=13, npz=20, k_split=2, n_split=2, bdt=900.0, moist_kappa=True, tol=1e-12):
    """A whole nonhydrostatic fv_dynamics call with use_cond (+ moist_kappa) on the six faces, from the JW temperature: T ->
    theta_m with moist_cv (fv_dynamics.F90:305-317, :381-388), q_con through d_sw and into both Riemann solvers, its halo (and
    cappa's) across the cube edges (dyn_core.F90:825, fv_dynamics.F90:464-465), the moist remap, back to T on the last step --
    FvDynamics.step_from_temperature over MultiContext against the six-face orchestration of the oracle."""
    import parity_nh as N
    import parity_remap as R
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, RDGAS
    from gfdl_atmos_cubed_sphere_amd.test_cases import set_eta
    cs, gs = CC.sphere(npx)
    if npz in (79, 127):
        ak, bk, ks, ptop = set_eta(npz)
    else:
        sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
        ak, bk = 300.0 * (1.0 - sig), sig.copy()
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=False, rdgas=L.RDGAS, grav=L.GRAV)
    CC.exchange(cs, st, ("phis",), "A")
    fl = DynFlags(n_split=n_split, hydrostatic=False, d_ext=0.0, ptop=float(ak[0]), use_cond=True, moist_kappa=moist_kappa)
    bd = gs[0].bd
    ng, nx = bd.ng, bd.nx
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    nq = 7
    mp = dict(R.MOIST6, sphum=1)
    zvir = 0.6077
    q0 = tracer_fields(cs, npz, nq)
    for q in q0:                     # water species: vapour ~1 %, condensates ~0.1 %
        q[..., 0] *= 0.01
        for iq in range(1, 6):
            q[..., iq] *= 0.001 / (1.0 + iq)
    # ---- oracle side: the conversion in numpy, then the six-face loop ----
    ost = []
    for t in range(6):
        s = st[t]
        T, dpc = s["pt"], s["delp"][c]
        dp1 = zvir * q0[t][c + (slice(None), 0)]
        q_con, cappa = bd.zeros("A", npz), bd.zeros("A", npz)
        cvm, qc = N.np_moist_cv(q0[t][c], mp, CP_AIR - RDGAS)
        q_con[c] = qc
        if moist_kappa:
            cappa[c] = RDGAS / (RDGAS + cvm / (1.0 + dp1))
            pkz = O.fexp(cappa[c] * O.flog((-RDGAS / GRAV) * dpc * T[c] * (1.0 + dp1) * (1.0 - qc) / s["delz"]))
        else:
            pkz = O.fexp(fl.akap * O.flog((-RDGAS / GRAV) * dpc * T[c] * (1.0 + dp1) / s["delz"]))
        th = T.copy(order="F")
        th[c] = T[c] * (1.0 + dp1) * (1.0 - qc) / pkz
        o = dict(s, pt=th, q_con=q_con)
        if moist_kappa:
            o["cappa"] = cappa
        ost.append(o)
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = {}
    try:
        fv = FvDynamics(mctx, fl, ak, bk, nq=nq, k_split=k_split, adiabatic=False, moist=mp, c2l_ord=2,
                        halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
        opar = dict(fv.remap_par, **dict(mp, moist_kappa=int(moist_kappa), use_cond=1))
        opar.pop("sphum")
        opar["sphum"] = 1
        dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
        ref = CC.oracle_fv_step_nh(cs, gs, fl, dp_ref, ost, ak, bk, bdt, k_split, opar, npz, q=q0, last_step=True)
        fv.dc.set_state([s["u"] for s in st], [s["v"] for s in st], [s["w"] for s in st], [s["delp"] for s in st],
                        [s["pt"] for s in st], [s["delz"] for s in st], [s["phis"] for s in st])
        fv.set_tracers(q0)
        if not moist_kappa:          # use_cond alone: the library is told q_con (the reference keeps it from the previous step)
            fv.dc.d["q_con"].upload([o["q_con"] for o in ost])
        fv.step_from_temperature(bdt)
        d = fv.dc.d
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r), ("w", "A", r)):
            got = d[n].download()
            for t in range(6):
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", bd.view(got[t], kind, *rr), bd.view(ref[t][n], kind, *rr), tol))
        got = d["q"].download()
        for t in range(6):
            for iq in (0, 1, 5, 6):
                worst["q"] = max(worst.get("q", 0.0), P.assert_close(f"face {t + 1} q{iq}", bd.view(got[t][:, :, :, iq], "A", *r),
                                                                       bd.view(ref[t]["q"][:, :, :, iq], "A", *r), tol))
        Tn = np.concatenate([bd.view(x, "A", *r).ravel() for x in d["pt"].download()])
        assert 150.0 < Tn.min() and Tn.max() < 400.0          # pt is a temperature again
    finally:
        mctx.close()
    return worst


def check_rayleigh_super(lib, npx=13, npz=20, hydrostatic=False, ideal=False, tau=5.0, rf_cutoff=80.e2):
    """Rayleigh_Super (fv_dynamics.F90:953-1124) -- what fv_dynamics calls for tau > 0 on the cubed sphere (:362-366) -- through the
    host's own dispatch (FvDynamics.rayleigh_friction over the six faces: cubed_to_latlon ord 2, then fv3_rayleigh_super) against the
    oracle's restatement on every face; ideal: the relaxation towards the winds of the first call (is_ideal_case), two calls"""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    cs, gs, st = CC.global_state(npx, npz, hydrostatic=hydrostatic)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = 300.0 * (1.0 - sig), sig.copy()
    fl = DynFlags(n_split=1, hydrostatic=hydrostatic, ptop=300.0, is_ideal_case=ideal)
    bd = gs[0].bd
    w0 = [None if hydrostatic else np.asfortranarray(s["w"] * 10.0) for s in st]
    mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
    worst = 0.0
    try:
        fv = FvDynamics(mctx, fl, ak, bk, tau=tau, rf_cutoff=rf_cutoff, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
        rf, pm, kmax = fv.rayleigh_profile(225.0)
        assert 0 < kmax < npz
        z = w0 if not hydrostatic else [np.zeros_like(s["delp"]) for s in st]
        dz = [bd.zeros("CC", npz) - 300.0 for _ in st]
        fv.dc.set_state([s["u"] for s in st], [s["v"] for s in st], z, [s["delp"] for s in st], [s["pt"] for s in st], dz,
                        [bd.zeros("A") for _ in st])
        ref = []
        for t in range(6):
            f = {k: st[t][k].copy(order="F") for k in ("u", "v", "pt")}
            f["w"] = None if hydrostatic else w0[t].copy(order="F")
            f["u00"], f["v00"] = (f["u"].copy(order="F"), f["v"].copy(order="F")) if ideal else (None, None)
            ref.append(f)
        for rep in range(2 if ideal else 1):
            fv.rayleigh_friction(225.0)
            for t in range(6):
                f = ref[t]
                ua, va = bd.zeros("A", npz), bd.zeros("A", npz)
                O.c2l(gs[t], npz, 2, f["u"], f["v"], ua, va)
                O.rayleigh_super(gs[t], kmax, not ideal, hydrostatic, fl.cp_air, fl.rdgas, fl.ptop, pm[:kmax], rf[:kmax], ua, va, f["pt"],
                                 f["u"], f["v"], f["w"], f["u00"], f["v00"])
            if ideal and rep == 0:          # move the winds away from u00 so that the second call relaxes something
                for t in range(6):
                    ref[t]["u"] *= 1.25
                    ref[t]["v"] *= 0.75
                fv.dc.d["u"].upload([r_["u"] for r_ in ref])
                fv.dc.d["v"].upload([r_["v"] for r_ in ref])
        d = fv.dc.d
        i0, i1, j0, j1 = bd.is_, bd.ie, bd.js, bd.je
        for n, kind, r in (("u", "U", (i0, i1, j0, j1 + 1)), ("v", "V", (i0, i1 + 1, j0, j1)), ("pt", "A", (i0, i1, j0, j1))) + \
                (() if hydrostatic else (("w", "A", (i0, i1, j0, j1)),)):
            got = d[n].download()
            for t in range(6):
                worst = max(worst, P.assert_close(f"face {t + 1} {n}", bd.view(got[t], kind, *r), bd.view(ref[t][n], kind, *r), 1e-14))
        assert np.max(np.abs(ref[0]["u"][:, :, 0] - st[0]["u"][:, :, 0])) > 1e-5 * np.max(np.abs(st[0]["u"][:, :, 0]))
    finally:
        mctx.close()
    return worst


def check_jw_consv(lib, npx=13, npz=20, k_split=2, n_split=2, bdt=900.0, consv_te=1.0, tol=1e-12, face=None, dist=None):
    """consv_te on the whole sphere (BASELINE config 2's kind of run: hydrostatic JW): compute_total_energy of the six faces before
    the loop, the energy fixer after the last remap with its two area-weighted global sums over the sphere, step 9a with dtmp --
    FvDynamics.step_from_temperature over MultiContext against the oracle's restatements and math.fsum over the six faces"""
    import math
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    cs, gs = CC.sphere(npx)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = 300.0 * (1.0 - sig), sig.copy()
    st = cs.jablonowski_williamson(ak, bk, hydrostatic=True, rdgas=L.RDGAS, grav=L.GRAV)
    CC.exchange(cs, st, ("phis",), "A")
    fl = DynFlags(n_split=n_split, hydrostatic=True, d_ext=0.0, ptop=float(ak[0]))
    bd = gs[0].bd
    ng, nx = bd.ng, bd.nx
    c = (slice(ng, ng + nx), slice(ng, ng + nx))
    pes, pelns, pkzs, ost = [], [], [], []
    for s_ in st:                     # what p_var left in the state: pe, peln, pkz of the hydrostatic column
        pe3 = ak[0] + np.concatenate([np.zeros((nx, nx, 1)), np.cumsum(s_["delp"][c], axis=2)], axis=2)
        peln3 = O.flog(pe3).reshape(pe3.shape)
        pk3 = O.fexp(fl.akap * peln3).reshape(pe3.shape)
        pkz = np.asfortranarray((pk3[:, :, 1:] - pk3[:, :, :-1]) / (fl.akap * (peln3[:, :, 1:] - peln3[:, :, :-1])))
        pe = np.zeros((nx + 2, npz + 1, nx + 2), order="F")
        pe[1:-1, :, 1:-1] = np.transpose(pe3, (0, 2, 1))
        pes.append(pe)
        pelns.append(np.asfortranarray(np.transpose(peln3, (0, 2, 1))))
        pkzs.append(pkz)
        th = s_["pt"].copy(order="F")
        th[c] = s_["pt"][c] / pkz
        ost.append(dict(s_, pt=th))
    if face is None:
        mctx = MultiContext([Context(g, npz, lib=lib) for g in gs])
        halo = CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx))
    else:       # one face per rank: the two global sums of the fixer travel as integer digits through dist.all_reduce
        from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeRankAdapter
        mctx = Context(gs[face], npz, lib=lib)
        halo = CubeRankAdapter(mctx, face, npx, dist, topo=CC.product_topo(npx))
    pick = (lambda xs: xs) if face is None else (lambda xs: xs[face])
    worst = {}
    try:
        fv = FvDynamics(mctx, fl, ak, bk, k_split=k_split, c2l_ord=2, consv_te=consv_te, halo=halo, dist=dist)
        par = dict(fv.remap_par)
        areas = [np.asarray(g.m["area"])[c] for g in gs]
        te0 = [bd.zeros("CC") for _ in gs]
        for t in range(6):
            O.compute_total_energy(gs[t], npz, par, False, st[t]["u"], st[t]["v"], None, None, st[t]["pt"], st[t]["delp"], None, None,
                                   pes[t], pelns[t], st[t]["phis"], te0[t])
        ref = CC.oracle_fv_step_hydro(cs, gs, fl, ost, ak, bk, bdt, k_split, par, npz, last_step=2)
        te2, z1, z0 = ([bd.zeros("CC") for _ in gs] for _ in range(3))
        for t in range(6):
            x = ref[t]
            O.energy_fixer_sums(gs[t], npz, par, False, x["u"], x["v"], None, None, x["pt"], x["delp"], None, x["pe"], x["peln"],
                                st[t]["phis"], x["pkz"], x["pk"], te0[t], te2[t], z1[t], z0[t])
        dtmp = consv_te * math.fsum(np.concatenate([(a * b).ravel() for a, b in zip(te2, areas)])) / \
            math.fsum(np.concatenate([(a * b).ravel() for a, b in zip(z0, areas)]))
        for t in range(6):
            O.remap_finish(gs[t], npz, par, dtmp, ref[t]["pt"], ref[t]["pkz"], None)
        z = [np.zeros_like(s_["delp"]) for s_ in st]
        fv.dc.set_state(pick([s_["u"] for s_ in st]), pick([s_["v"] for s_ in st]), pick(z), pick([s_["delp"] for s_ in st]),
                        pick([s_["pt"] for s_ in st]), pick([bd.zeros("CC", npz) for _ in st]), pick([s_["phis"] for s_ in st]))
        fv.dc.d["pe"].upload(pick(pes))
        fv.dc.d["peln"].upload(pick(pelns))
        fv.dc.d["pkz"].upload(pick(pkzs))
        fv.step_from_temperature(bdt)
        worst["dtmp"] = abs(fv.dtmp - dtmp) / abs(dtmp)
        assert worst["dtmp"] < 1e-11 and abs(dtmp) > 1e-14, (fv.dtmp, dtmp)
        d = fv.dc.d
        r = (bd.is_, bd.ie, bd.js, bd.je)
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)), ("delp", "A", r),
                            ("pt", "A", r)):
            got = d[n].download()
            for t in (range(6) if face is None else (face,)):
                g_ = got[t] if face is None else got
                worst[n] = max(worst.get(n, 0.0), P.assert_close(f"face {t + 1} {n}", bd.view(g_, kind, *rr), bd.view(ref[t][n], kind, *rr), tol))
    finally:
        mctx.close()
    return worst


def check_adv_pe(lib, npx=13, npz=6, faces=range(6)):
    """adv_pe (dyn_core.F90:1529-1632) on the faces: pem from a delp with its halo exchanged across the cube edges, the corner
    pressures by the cubed a2b_ord2 (edges, corners), the projection on ec1 / ec2 / en1 / en2 -- oracle vs fv3_adv_pe; the term
    vanishes for a horizontally uniform pressure"""
    cs, gs, st = CC.global_state(npx, npz, hydrostatic=True)
    CC.exchange(cs, st, ("delp",), "A")
    bd = gs[0].bd
    r = (bd.is_, bd.ie, bd.js, bd.je)
    rng = np.random.default_rng(17)
    worst = 0.0
    for t in faces:
        g = gs[t]
        a3 = cs.grids[t]["agrid3"]
        ua = np.asfortranarray(np.stack([20.0 * np.sin(2.0 * a3[..., 0] + 0.3 * k) + 5.0 * a3[..., 2] for k in range(npz)], axis=-1))
        va = np.asfortranarray(np.stack([15.0 * np.cos(3.0 * a3[..., 1] - 0.2 * k) for k in range(npz)], axis=-1))
        om0 = np.asfortranarray(rng.uniform(-1.0, 1.0, bd.shape("A", npz)))
        ref = om0.copy(order="F")
        O.adv_pe(g, npz, 300.0, ua, va, st[t]["delp"], ref)
        assert np.max(np.abs(bd.view(ref, "A", *r) - bd.view(om0, "A", *r))) > 1e-6
        flat = om0.copy(order="F")
        O.adv_pe(g, npz, 300.0, ua, va, np.asfortranarray(np.full_like(st[t]["delp"], 1000.0)), flat)
        # uniform pressure: the closed line integral of p n dl around a cell is p * (sum of the edge vectors) -- zero to the
        # curvature of the cell (O(h^2): 2.2e-5, 5.7e-6, 1.5e-6 at C12, C24, C48 against a term of 3e-3), far below the term itself
        assert np.max(np.abs(bd.view(flat, "A", *r) - bd.view(om0, "A", *r))) < 2e-2 * np.max(np.abs(bd.view(ref, "A", *r) - bd.view(om0, "A", *r)))
        ctx = Context(g, npz, lib=lib)
        try:
            d_om = ctx.from_host(om0)
            ctx.adv_pe(300.0, ctx.from_host(ua), ctx.from_host(va), ctx.from_host(st[t]["delp"]), d_om)
            worst = max(worst, P.assert_close(f"face {t + 1} omga", bd.view(d_om.download(), "A", *r), bd.view(ref, "A", *r), 1e-14))
        finally:
            ctx.close()
    return worst


def check_face_group(lib, npx=13, npz=5, n_split=2, bdt=300.0, hydrostatic=False, flags=None):
    """fv3_group (include/fv3_mi355x.h): the acoustic substep loop of the whole sphere with the six contexts as ONE group -- every
    kernel the faces issue in turn runs as one launch over all of them -- against the same loop with six separate launches per kernel:
    bit for bit the same fields; and the group did merge (nearly every launch ran all six faces at once)."""
    from gfdl_atmos_cubed_sphere_amd.cubed_dyn import CubeHaloAdapter, MultiContext
    from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
    cs, gs, st = CC.hydro_state(npx, npz) if hydrostatic else CC.nh_state(npx, npz)
    fl = DynFlags(n_split=n_split, hydrostatic=hydrostatic, **(flags or {}))
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    dp0 = np.diff(fl.ptop + (1.0e5 - fl.ptop) * sig)
    out, stats = [], None
    for group in (False, True):
        mctx = MultiContext([Context(g, npz, lib=lib) for g in gs], group=group)
        try:
            assert (mctx.group is not None) == group
            dc = DynCore(mctx, fl, dp0, halo=CubeHaloAdapter(mctx, npx, topo=CC.product_topo(npx)))
            bd = gs[0].bd
            zero = [np.zeros_like(s["delp"]) for s in st]
            dc.set_state([s["u"] for s in st], [s["v"] for s in st], zero if hydrostatic else [s["w"] for s in st],
                         [s["delp"] for s in st], [s["pt"] for s in st],
                         [bd.zeros("CC", npz) for _ in st] if hydrostatic else [s["delz"] for s in st], [s["phis"] for s in st])
            if group:
                mctx.group.stats()
            dc.run(bdt)
            if group:
                mctx.flush()
                stats = mctx.group.stats()
            names = ("u", "v", "delp", "pt", "mfx", "mfy", "cx", "cy", "pk") + (() if hydrostatic else ("w", "zh", "delz"))
            out.append({n: dc.d[n].download() for n in names})
        finally:
            mctx.close()
    for n in out[0]:
        for t in range(6):
            assert np.array_equal(out[0][n][t], out[1][n][t]), f"face group: {n} of face {t + 1} differs from the separate launches"
    merged, single = stats
    assert merged > 0 and single <= 0.05 * merged, f"face group: {merged} merged launches, {single} single ones"
    return stats
