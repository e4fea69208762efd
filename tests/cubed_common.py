"""Cubed-sphere test harness: a global smooth state on the six faces (one tile per face), the single-process emulation of
the reference's halo updates (gfdl_atmos_cubed_sphere_amd.cubed_sphere.CubeTopology.update) and the oracle's c_sw -> d_sw
driven face by face the way dyn_core drives them (dyn_core.F90:439-447, :565-578, :762-772)."""
from __future__ import annotations

import numpy as np

import grid_oracle as GO
import oracle_lib as O
from grid_oracle import level_coefficients           # model/dyn_core.F90:666-733 as the ORACLE restates it (oracle/fv_grid.c)
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags                    # the flag container only
from gfdl_atmos_cubed_sphere_amd.synthetic import CSW_OUT, DSW_PAR           # field-name / default-parameter lists only

F = np.asfortranarray
_CACHE = {}
_PTOPO = {}


def _unit(v):
    return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


def _mid(p, q):
    return _unit(p + q)


def sphere(npx):
    """(cs, gs): the six tiles as the ORACLE builds them (tests/grid_oracle.py over oracle/fv_grid.c -- geometry, metric terms,
    halo topology, nothing from the product's cubed_sphere.py) and their gridstructs.  The parity tests hand the SAME
    gridstructs to the product kernels (as a real integration hands them the reference's init_grid output); the product's own
    numpy geometry is held to this one by tests/test_grid_oracle.py."""
    if npx not in _CACHE:
        cs = GO.ref_sphere(npx)
        _CACHE[npx] = (cs, [cs.gridstruct(t) for t in range(6)])
    return _CACHE[npx]


def product_topo(npx):
    """the PRODUCT's halo tables (what its device gathers / message packs are built from): used for the product side of a parity
    test only; tests/test_grid_oracle.py requires them to equal the oracle's row for row"""
    if npx not in _PTOPO:
        from gfdl_atmos_cubed_sphere_amd.cubed_sphere import CubeTopology
        _PTOPO[npx] = CubeTopology(npx)
    return _PTOPO[npx]


def wind(p, strength=30.0):
    """a smooth tangent wind field on the unit sphere (m/s): solid-body rotation about a tilted axis + a wavy part"""
    axis = _unit(np.array([0.3, -0.5, 0.8]))
    v = strength * np.cross(axis, p)
    v = v + 0.4 * strength * np.cross(np.array([1.0, 0.0, 0.0]), p) * np.sin(3.0 * p[..., 2:3] + 2.0 * p[..., 1:2])
    return v


def _ripple(p):
    return np.sin(37.0 * p[..., 0] + 11.0 * p[..., 1]) * np.cos(29.0 * p[..., 2] - 17.0 * p[..., 1]) + 0.5 * np.sin(53.0 * p[..., 1] * p[..., 0])


def scalar(p, k, npz, base, amp):
    return base * (1.0 + amp * (np.sin(2.0 * p[..., 0] + 0.3 * k) * np.cos(3.0 * p[..., 1]) + 0.5 * p[..., 2] ** 2) + 0.1 * k / max(npz, 1))


def global_state(npx, npz, hydrostatic=False, seed=3, noise=0.02):
    """u, v (D grid, covariant components along the cell edges), delp, pt, w per face, halos filled"""
    cs, gs = sphere(npx)
    st = [dict() for _ in range(6)]
    for t in range(6):
        g3, a3 = cs.grids[t]["grid3"], cs.grids[t]["agrid3"]
        tx, mx = _unit(g3[1:, :] - g3[:-1, :]), _mid(g3[1:, :], g3[:-1, :])
        ty, my = _unit(g3[:, 1:] - g3[:, :-1]), _mid(g3[:, 1:], g3[:, :-1])
        u = np.stack([np.sum(wind(mx) * tx, -1) * (1.0 + 0.05 * k / npz) for k in range(npz)], axis=-1)
        v = np.stack([np.sum(wind(my) * ty, -1) * (1.0 + 0.05 * k / npz) for k in range(npz)], axis=-1)
        # small-scale structure as a function of POSITION (the edge points of two faces are the same physical points and
        # must carry the same values: the model keeps them equal, dyn_core.F90:1151-1163)
        u = u * (1.0 + noise * _ripple(mx)[..., None])
        v = v * (1.0 + noise * _ripple(my)[..., None])
        st[t]["u"], st[t]["v"] = F(u), F(v)
        st[t]["delp"] = F(np.stack([scalar(a3, k, npz, 800.0, 0.1) for k in range(npz)], axis=-1) * (1.0 + 0.2 * noise * _ripple(a3)[..., None]))
        st[t]["pt"] = F(np.stack([scalar(a3, k + 2, npz, 300.0, 0.05) for k in range(npz)], axis=-1) * (1.0 + 0.2 * noise * _ripple(a3 * 1.1)[..., None]))
        if not hydrostatic:
            st[t]["w"] = F(np.stack([scalar(a3, k + 5, npz, 0.5, 1.0) - 0.5 for k in range(npz)], axis=-1))
    exchange(cs, st, ("delp", "pt") + (() if hydrostatic else ("w",)), "A")
    exchange_pair(cs, st, "u", "v", "D")
    return cs, gs, st


def exchange(cs, st, names, kind):
    for n in names:
        cs.topo.update(kind, [s[n] for s in st])


def exchange_pair(cs, st, a, b, kind, vector=True):
    cs.topo.update(kind, ([s[a] for s in st], [s[b] for s in st]), vector=vector)


def oracle_c_sw(gs, st, npz, dt2, hydrostatic, nord=1):
    out = []
    for t in range(6):
        bd = gs[t].bd
        f = {k: v.copy(order="F") for k, v in st[t].items()}
        for n, kind in CSW_OUT:
            f[n] = bd.zeros(kind, npz)
        O.c_sw_3d(gs[t], npz, f, nord=nord, dt2=dt2, hydrostatic=hydrostatic)
        out.append(f)
    return out


def dsw_work_arrays(bd, npz):
    f = {}
    for n, kind in (("mfx", "FX"), ("mfy", "FY"), ("cx", "CX"), ("cy", "CY"), ("crx", "CX"), ("cry", "CY"), ("xfx", "CX"), ("yfx", "CY"),
                    ("heat_source", "CC"), ("diss_est", "CC")):
        f[n] = bd.zeros(kind, npz)
    return f


def oracle_pair(npx, npz, dt=300.0, hydrostatic=False, par_over=None, flags=None, st=None, use_cond=False):
    """c_sw on every face, the halo updates dyn_core does in between, d_sw on every face.  Returns (cs, gs, before, after)."""
    if st is None:
        cs, gs, st = global_state(npx, npz, hydrostatic)
    else:
        cs, gs = sphere(npx)
    if use_cond:      # a condensate mixing ratio in [0, 0.02] as a function of position (halo by exchange)
        for t in range(6):
            a3 = cs.grids[t]["agrid3"]
            st[t]["q_con"] = F(np.stack([0.01 * (1.0 + np.sin(3.0 * a3[..., 0] + 0.4 * k) * np.cos(2.0 * a3[..., 1])) for k in range(npz)], axis=-1))
        exchange(cs, st, ("q_con",), "A")
    fl = DynFlags(**(flags or {}))
    c = oracle_c_sw(gs, st, npz, 0.5 * dt, hydrostatic, nord=fl.nord)
    exchange_pair(cs, c, "uc", "vc", "C")                    # dyn_core.F90:565 (CGRID_NE)
    if fl.nord > 0:
        exchange(cs, c, ("divg_d",), "B")                    # :451 (position = CORNER)
    par = dict(DSW_PAR)
    par.update(dt=dt, nord=1, nord_v=1, nord_w=1, nord_t=1, d2_bg=0., damp_v=0., damp_w=0., damp_t=0., d_con=0.,
               hydrostatic=int(hydrostatic), use_cond=0)
    par.update(par_over or {})
    if use_cond:
        par["use_cond"] = 1
        for t in range(6):
            c[t]["q_con"] = st[t]["q_con"].copy(order="F")
    lev = level_coefficients(npz, fl)
    before = [{k: v.copy(order="F") for k, v in f.items()} for f in c]
    for t in range(6):
        c[t].update(dsw_work_arrays(gs[t].bd, npz))
        O.d_sw_3d(gs[t], npz, par, lev, c[t])
    return cs, gs, before, c


def hydro_state(npx, npz, ptop=300.0):
    """u, v, delp, pt (theta), phis per face for the hydrostatic substep loop: the winds of global_state, a smooth surface
    pressure and potential temperature as functions of position"""
    cs, gs, st = global_state(npx, npz, hydrostatic=True)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    out = []
    for t in range(6):
        a3 = cs.grids[t]["agrid3"]
        ps = 1.0e5 * (1.0 + 0.01 * (np.sin(2.0 * a3[..., 0] + 1.0) * np.cos(3.0 * a3[..., 1]) + 0.5 * a3[..., 2]))
        pe = ptop + (ps[:, :, None] - ptop) * sig[None, None, :]
        delp = F(np.diff(pe, axis=2))
        pm = delp / np.log(pe[:, :, 1:] / pe[:, :, :-1])
        T = 300.0 - 60.0 * (1.0 - sig[None, None, 1:]) + 2.0 * np.sin(3.0 * a3[..., 1:2] + a3[..., 2:3])
        pt = F(T * pm ** (-2.0 / 7.0))
        phis = F(9.80665 * 200.0 * (1.0 + np.sin(2.0 * a3[..., 0]) * np.cos(2.0 * a3[..., 1] + 1.0)))
        out.append(dict(u=st[t]["u"], v=st[t]["v"], delp=delp, pt=pt, phis=phis))
    return cs, gs, out


def _n_con(fl, npz):
    if fl.convert_ke or (fl.do_vort_damp and fl.vtdm4 > 1.0e-4):
        return npz
    if fl.d2_bg_k1 < 1.0e-3:
        return 0
    return 1 if fl.d2_bg_k2 < 1.0e-3 else 2


def _oracle_heating(cs, gs, fl, f, npz, bdt, hydrostatic):
    """dyn_core.F90:1300-1355 on six faces: halo update of the accumulated heat source, del2_cubed, the heating of pt"""
    n_con = _n_con(fl, npz)
    if n_con == 0 or not fl.d_con > 1.0e-5:
        return
    exchange(cs, f, ("heat_source",), "A")
    for t in range(6):
        x = f[t]
        O.del2_cubed(gs[t], npz, 0.20 * gs[t].da_min, min(3, fl.nord + 1), x["heat_source"])
        O.apply_heat_source(gs[t], npz, n_con, hydrostatic, bdt, fl.delt_max, fl.cp_air, fl.cp_air - fl.rdgas, fl.rdgas, fl.grav,
                            x["pt"], x["heat_source"], x["delp"], x["pkz"] if hydrostatic else x["delz"], x["pkz"],
                            x["cappa"] if (fl.moist_kappa and not hydrostatic) else None)          # :1338-1340


def oracle_substeps_hydro(cs, gs, fl, st, bdt, npz):
    """the hydrostatic substep loop of dyn_core (dyn_core.F90:313-1286) over the oracle's routines on six
    faces, with the halo updates where dyn_core has them -- the six-face twin of oracle_dyn_core.run_hydrostatic"""
    f = [{k: F(v.copy()) for k, v in s.items()} for s in st]
    bd = gs[0].bd
    nx, ny = bd.nx, bd.ny
    for t in range(6):
        for n, kind, nk in (("delpc", "A", npz), ("ptc", "A", npz), ("uc", "V", npz), ("vc", "U", npz), ("ua", "A", npz),
                            ("va", "A", npz), ("ut", "A", npz), ("vt", "A", npz), ("divgd", "B", npz), ("gz", "A", npz + 1),
                            ("pkc", "A", npz + 1), ("crx", "CX", npz), ("xfx", "CX", npz), ("cry", "CY", npz), ("yfx", "CY", npz),
                            ("mfx", "FX", npz), ("mfy", "FY", npz), ("cx", "CX", npz), ("cy", "CY", npz), ("heat_s", "CC", npz),
                            ("diss_e", "CC", npz), ("pk", "CC", npz + 1), ("pkz", "CC", npz)):
            f[t][n] = bd.zeros(kind, nk)
        f[t]["divg2"] = bd.zeros("A")
        f[t]["heat_source"] = bd.zeros("A", npz)
        f[t]["pe"] = np.zeros((nx + 2, npz + 1, ny + 2), order="F")
        f[t]["peln"] = np.zeros((nx, npz + 1, ny), order="F")
    lev = level_coefficients(npz, fl)
    heating = fl.d_con > 1.0e-5
    n_split = fl.n_split
    dt = bdt / float(n_split)
    dt2 = 0.5 * dt
    ptk = fl.ptop ** fl.akap
    par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm, hord_dp=fl.hord_dp, nord=1,
               nord_v=1, nord_w=1, nord_t=1, dddmp=fl.dddmp, d2_bg=0.0, d4_bg=fl.d4_bg, damp_v=0.0, damp_w=0.0, damp_t=0.0, d_con=0.0,
               kgb=fl.ke_bg, hydrostatic=1, use_cond=0)
    exchange(cs, f, ("delp", "pt"), "A")
    exchange_pair(cs, f, "u", "v", "D")
    i0 = j0 = bd.ng
    for it in range(1, n_split + 1):
        for t in range(6):
            x = f[t]
            cs_ = dict(delpc=x["delpc"], delp=x["delp"], ptc=x["ptc"], pt=x["pt"], u=x["u"], v=x["v"], uc=x["uc"], vc=x["vc"], ua=x["ua"],
                       va=x["va"], ut=x["ut"], vt=x["vt"], divg_d=x["divgd"])
            O.c_sw_3d(gs[t], npz, cs_, nord=fl.nord, dt2=dt2, hydrostatic=True)
        if fl.nord > 0:
            exchange(cs, f, ("divgd",), "B")
        for t in range(6):
            x = f[t]
            O.geopk(gs[t], npz, fl.ptop, fl.akap, fl.cp_air, x["pe"], x["peln"], x["delpc"], x["pkc"], x["gz"], x["phis"], x["ptc"], x["pkz"], True)
            O.p_grad_c(gs[t], npz, dt2, x["delpc"], x["pkc"], x["gz"], x["uc"], x["vc"], True)
        exchange_pair(cs, f, "uc", "vc", "C")
        if fl.inline_q and "q" in f[0]:
            exchange(cs, f, ("q",), "A")                                       # dyn_core.F90:341 / :573 (pack 10)
        for t in range(6):
            x = f[t]
            ds = dict(delpc=x["vt"], delp=x["delp"], ptc=x["ptc"], pt=x["pt"], u=x["u"], v=x["v"], uc=x["uc"], vc=x["vc"], ua=x["ua"],
                      va=x["va"], divg_d=x["divgd"], mfx=x["mfx"], mfy=x["mfy"], cx=x["cx"], cy=x["cy"], crx=x["crx"], cry=x["cry"],
                      xfx=x["xfx"], yfx=x["yfx"], heat_source=x["heat_s"], diss_est=x["diss_e"])
            delp_old = x["delp"].copy(order="F")
            if fl.inline_q and "q" in x:                                     # sw_core.F90:1020-1043
                ds["inline_q"] = x["q"]
            O.d_sw_3d(gs[t], npz, par, lev, ds)
            if heating:
                x["heat_source"][i0:i0 + nx, j0:j0 + ny, :] += x["heat_s"]
            O.divg2_ext(gs[t], npz, fl.d_ext, delp_old, x["vt"], x["divg2"])      # dyn_core.F90:745-747, :791-848
        exchange(cs, f, ("delp", "pt"), "A")
        for t in range(6):
            x = f[t]
            O.geopk(gs[t], npz, fl.ptop, fl.akap, fl.cp_air, x["pe"], x["peln"], x["delp"], x["pkc"], x["gz"], x["phis"], x["pt"], x["pkz"], False)
            if it == n_split:
                x["pk"][...] = x["pkc"][i0:i0 + nx, j0:j0 + ny, :]
            if fl.beta > 0.0:                                              # dyn_core.F90:1018-1019
                for n, kind in (("du", "U"), ("dv", "V")):
                    x.setdefault(n, bd.zeros(kind, npz))
                O.grad1_p_update(gs[t], npz, x["divg2"], x["u"], x["v"], x["pkc"], x["gz"], dt, ptk, 0.0 if it == 1 else fl.beta,
                                 x["du"], x["dv"])
            else:
                O.one_grad_p_hydro(gs[t], npz, dt, ptk, x["divg2"], x["u"], x["v"], x["pkc"], x["gz"])
        if it != n_split:
            exchange_pair(cs, f, "u", "v", "D")
        else:
            exchange_pair(cs, f, "u", "v", "Dedge")          # mpp_get_boundary, dyn_core.F90:1151-1163
    _oracle_heating(cs, gs, fl, f, npz, bdt, True)
    return f


def nh_state(npx, npz, ptop=300.0):
    """the state of hydro_state + w and a hydrostatically balanced delz (pt = T / p**kappa: delz = -rd/g * pt * pm**kappa * d(ln p)), for the nonhydrostatic substep loop"""
    cs, gs, st = hydro_state(npx, npz, ptop)
    bd = gs[0].bd
    c = (slice(bd.ng, bd.ng + bd.nx), slice(bd.ng, bd.ng + bd.ny))
    kap, rd, grav = 2.0 / 7.0, 287.05, 9.80665
    for t in range(6):
        a3 = cs.grids[t]["agrid3"]
        x = st[t]
        pe = ptop + np.concatenate([np.zeros(x["delp"].shape[:2] + (1,)), np.cumsum(x["delp"], axis=2)], axis=2)
        peln = np.log(pe)
        pm = x["delp"] / (peln[:, :, 1:] - peln[:, :, :-1])      # the layer-mean pressure the Riemann solver recovers
        x["delz"] = F((-rd / grav * x["pt"] * pm ** kap * (peln[:, :, 1:] - peln[:, :, :-1]))[c])
        x["w"] = F(np.stack([0.2 * (scalar(a3, k + 5, npz, 0.5, 1.0) - 0.5) for k in range(npz)], axis=-1))
    return cs, gs, st


def oracle_substeps_nh(cs, gs, fl, dp_ref, st, bdt, npz):
    """the nonhydrostatic substep loop of dyn_core (dyn_core.F90:313-1286, d_ext = 0, d_con = 0) over the oracle's routines on
    six faces with the halo updates where dyn_core has them -- the six-face twin of oracle_dyn_core.run"""
    f = [{k: F(v.copy()) for k, v in s.items()} for s in st]
    bd = gs[0].bd
    nx, ny, ng = bd.nx, bd.ny, bd.ng
    for t in range(6):
        for n, kind, nk in (("delpc", "A", npz), ("ptc", "A", npz), ("uc", "V", npz), ("vc", "U", npz), ("ua", "A", npz),
                            ("va", "A", npz), ("omga", "A", npz), ("ut", "A", npz), ("vt", "A", npz), ("divgd", "B", npz),
                            ("gz", "A", npz + 1), ("pkc", "A", npz + 1), ("zh", "A", npz + 1), ("pk3", "A", npz + 1),
                            ("crx", "CX", npz), ("xfx", "CX", npz), ("cry", "CY", npz), ("yfx", "CY", npz),
                            ("mfx", "FX", npz), ("mfy", "FY", npz), ("cx", "CX", npz), ("cy", "CY", npz),
                            ("heat_s", "CC", npz), ("diss_e", "CC", npz), ("pk", "CC", npz + 1)):
            f[t][n] = bd.zeros(kind, nk)
        f[t]["ws3"], f[t]["ws"] = bd.zeros("A"), bd.zeros("CC")
        f[t]["heat_source"] = bd.zeros("A", npz)
        f[t]["pkz"] = bd.zeros("CC", npz)
        f[t]["pe"] = np.zeros((nx + 2, npz + 1, ny + 2), order="F")
        f[t]["peln"] = np.zeros((nx, npz + 1, ny), order="F")
        f[t]["zs"] = F(f[t]["phis"] * (1.0 / fl.grav))
    lev = level_coefficients(npz, fl)
    cn = dict(grav=fl.grav, rdgas=fl.rdgas, cp_air=fl.cp_air, akap=fl.akap, ptop=fl.ptop, p_fac=fl.p_fac, a_imp=fl.a_imp)
    n_split = fl.n_split
    dt = bdt / float(n_split)
    dt2, rdt = 0.5 * dt, 1.0 / dt
    ptk, peln1 = fl.ptop ** fl.akap, np.log(fl.ptop)
    par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm, hord_dp=fl.hord_dp, nord=1,
               nord_v=1, nord_w=1, nord_t=1, dddmp=fl.dddmp, d2_bg=0.0, d4_bg=fl.d4_bg, damp_v=0.0, damp_w=0.0, damp_t=0.0, d_con=0.0,
               kgb=fl.ke_bg, hydrostatic=0, use_cond=int(fl.use_cond))
    qc = lambda x: x["q_con"] if fl.use_cond else None            # nh_core.F90:96-166, nh_utils.F90:383-438
    cap = lambda x: x["cappa"] if fl.moist_kappa else None
    ndif = np.concatenate([lev["nord_v"], lev["nord_v"][-1:]]).astype(np.int32)
    damp = np.concatenate([lev["damp_vt"], lev["damp_vt"][-1:]])
    exchange(cs, f, ("delp", "pt"), "A")
    exchange_pair(cs, f, "u", "v", "D")
    for it in range(1, n_split + 1):
        remap_step = it == n_split
        exchange(cs, f, ("w",), "A")
        for t in range(6):
            x = f[t]
            if it == 1:
                gz = x["gz"]
                gz[ng:ng + nx, ng:ng + ny, npz] = x["zs"][ng:ng + nx, ng:ng + ny]
                for k in range(npz - 1, -1, -1):
                    gz[ng:ng + nx, ng:ng + ny, k] = gz[ng:ng + nx, ng:ng + ny, k + 1] - x["delz"][:, :, k]
        if it == 1:
            exchange(cs, f, ("gz",), "A")
            for t in range(6):
                f[t]["zh"][...] = f[t]["gz"]
        else:
            for t in range(6):
                f[t]["gz"][...] = f[t]["zh"]
        for t in range(6):
            x = f[t]
            cs_ = dict(delpc=x["delpc"], delp=x["delp"], ptc=x["ptc"], pt=x["pt"], u=x["u"], v=x["v"], w=x["w"], uc=x["uc"], vc=x["vc"],
                       ua=x["ua"], va=x["va"], wc=x["omga"], ut=x["ut"], vt=x["vt"], divg_d=x["divgd"])
            O.c_sw_3d(gs[t], npz, cs_, nord=fl.nord, dt2=dt2, hydrostatic=False)
        if fl.nord > 0:
            exchange(cs, f, ("divgd",), "B")
        for t in range(6):
            x = f[t]
            O.update_dz_c(gs[t], npz, dt2, dp_ref, x["zs"], x["ut"], x["vt"], x["gz"], x["ws3"])
            O.riem_solver_c(gs[t], npz, dt2, cn, x["phis"], x["omga"], x["ptc"], x["delpc"], x["gz"], x["pkc"], x["ws3"], qc(x), cap(x))
            O.p_grad_c(gs[t], npz, dt2, x["delpc"], x["pkc"], x["gz"], x["uc"], x["vc"], False)
        exchange_pair(cs, f, "uc", "vc", "C")
        if fl.inline_q and "q" in f[0]:
            exchange(cs, f, ("q",), "A")                                       # dyn_core.F90:341 / :573 (pack 10)
        for t in range(6):
            x = f[t]
            ds = dict(delpc=x["vt"], delp=x["delp"], ptc=x["ptc"], pt=x["pt"], u=x["u"], v=x["v"], w=x["w"], uc=x["uc"], vc=x["vc"],
                      ua=x["ua"], va=x["va"], divg_d=x["divgd"], mfx=x["mfx"], mfy=x["mfy"], cx=x["cx"], cy=x["cy"], crx=x["crx"],
                      cry=x["cry"], xfx=x["xfx"], yfx=x["yfx"], heat_source=x["heat_s"], diss_est=x["diss_e"])
            if fl.use_cond:
                ds["q_con"] = x["q_con"]
            if fl.inline_q and "q" in x:                                     # sw_core.F90:1020-1043
                ds["inline_q"] = x["q"]
            delp_old = x["delp"].copy(order="F") if fl.beta < -0.1 else None
            O.d_sw_3d(gs[t], npz, par, lev, ds)
            if fl.d_con > 1.0e-5:
                x["heat_source"][ng:ng + nx, ng:ng + ny, :] += x["heat_s"]
            if fl.beta < -0.1:                                               # dyn_core.F90:745-747, :791-848 (zeros when d_ext = 0)
                x.setdefault("divg2", bd.zeros("A"))
                O.divg2_ext(gs[t], npz, fl.d_ext, delp_old, x["vt"], x["divg2"])
        exchange(cs, f, ("delp", "pt"), "A")
        if fl.use_cond:
            exchange(cs, f, ("q_con",), "A")                       # dyn_core.F90:825 / :852
        for t in range(6):
            x = f[t]
            O.update_dz_d(gs[t], npz, ndif.copy(), damp.copy(), fl.hord_tm, dp_ref, x["zs"], x["zh"], x["crx"], x["cry"], x["xfx"],
                          x["yfx"], x["ws"], rdt)
            O.riem_solver3(gs[t], npz, dt, cn, x["zs"], x["w"], x["delz"], x["pt"], x["delp"], x["zh"], x["pe"], x["pkc"], x["pk3"],
                           x["pk"], x["peln"], x["ws"], fl.use_logp, remap_step, fl.beta < -0.1, qc(x), cap(x))   # fp_out: dyn_core.F90:939
        exchange(cs, f, ("zh", "pkc"), "A")
        for t in range(6):
            x = f[t]
            if remap_step:
                O.pe_halo(gs[t], npz, fl.ptop, x["pe"], x["delp"])
            O.pk3_halo(gs[t], npz, fl.ptop, fl.akap, x["pk3"], x["delp"], fl.use_logp)
            i0, i1 = ng - 2, ng + nx + 2
            x["gz"][i0:i1, i0:i1, :] = x["zh"][i0:i1, i0:i1, :] * fl.grav
            if fl.beta > 0.0:                                              # dyn_core.F90:1027-1028, beta_d :398-406
                for n, kind in (("du", "U"), ("dv", "V")):
                    x.setdefault(n, bd.zeros(kind, npz))
                O.split_p_grad(gs[t], npz, x["u"], x["v"], x["pkc"], x["gz"], x["delp"], x["pk3"], 0.0 if it == 1 else fl.beta, dt,
                               peln1 if fl.use_logp else ptk, x["du"], x["dv"])
            elif fl.beta < -0.1:                                           # dyn_core.F90:1029-1030
                O.one_grad_p_nh(gs[t], npz, dt, fl.ptop, x["divg2"], x["u"], x["v"], x["pkc"], x["gz"], x["delp"])
            else:
                O.nh_p_grad(gs[t], npz, x["u"], x["v"], x["pkc"], x["gz"], x["delp"], x["pk3"], dt, peln1 if fl.use_logp else ptk)
        if it != n_split:
            exchange_pair(cs, f, "u", "v", "D")
        else:
            exchange_pair(cs, f, "u", "v", "Dedge")
    _oracle_heating(cs, gs, fl, f, npz, bdt, False)
    return f


def _oracle_fill2d(cs, gs, npz, q, delp, which):
    """fill2D after tracer_2d (fv_dynamics.F90:542-556) on six faces: qt = q delp area, its halo, the sign-change fluxes"""
    bd = gs[0].bd
    for iq in which:
        qi = [np.asfortranarray(q[t][:, :, :, iq]) for t in range(6)]
        qt = [bd.zeros("A", npz) for _ in range(6)]
        for t in range(6):
            O.fill2d_mass(gs[t], npz, qi[t], delp[t], qt[t])
        cs.topo.update("A", qt)
        for t in range(6):
            O.fill2d_apply(gs[t], npz, qt[t], delp[t], qi[t])
            q[t][:, :, :, iq] = qi[t]


def _oracle_tracers(cs, gs, fl, npz, q, dp1, f):
    nq = q[0].shape[3]
    oracle_tracer_2d(cs, gs, npz, nq, q, dp1, [x["mfx"] for x in f], [x["mfy"] for x in f], [x["cx"] for x in f],
                     [x["cy"] for x in f], fl.hord_tr, 0)


def oracle_fv_step_hydro(cs, gs, fl, st, ak, bk, bdt, k_split, remap_par, npz, q=None, last_step=0, fill2d=()):
    """hydrostatic k_split loop on six faces over the oracle: dyn_core substeps -> tracer_2d -> Lagrangian_to_Eulerian"""
    mdt = bdt / float(k_split)
    cur = [{k: s[k].copy(order="F") for k in ("u", "v", "delp", "pt", "phis")} for s in st]
    q = None if q is None else [x.copy(order="F") for x in q]
    out = None
    bd = gs[0].bd
    for n_map in range(1, k_split + 1):
        dp1 = [c["delp"].copy(order="F") for c in cur]
        if fl.inline_q and q is not None:
            for t in range(6):
                cur[t]["q"] = q[t]
        f = oracle_substeps_hydro(cs, gs, fl, cur, mdt, npz)
        if fl.inline_q and q is not None:
            q = [x["q"] for x in f]
        elif q is not None:
            _oracle_tracers(cs, gs, fl, npz, q, dp1, f)
            _oracle_fill2d(cs, gs, npz, q, [x["delp"] for x in f], fill2d if fl.hord_tr < 8 else ())
        out = []
        for t in range(6):
            x = f[t]
            rf = dict(ps=bd.zeros("A"), pe=x["pe"], delp=x["delp"], pkz=x["pkz"], pk=x["pk"], u=x["u"], v=x["v"], pt=x["pt"], peln=x["peln"],
                      omga=bd.zeros("A", npz))
            if q is not None:
                rf["q"] = q[t]
            if remap_par.get("remap_te"):
                rf["hs"], rf["te"] = st[t]["phis"], bd.zeros("A", npz)
            O.lagrangian_to_eulerian(gs[t], npz, dict(remap_par, last_step=(int(last_step) if n_map == k_split else 0)), rf, ak, bk)
            cur[t] = dict(u=rf["u"], v=rf["v"], delp=rf["delp"], pt=rf["pt"], phis=st[t]["phis"])
            cur[t].update({n: x[n] for n in ("du", "dv") if n in x})       # dyn_core's saved arrays (dyn_core.F90:278-283)
            out.append(dict(cur[t], pkz=rf["pkz"], ps=rf["ps"], pe=rf["pe"], peln=rf["peln"], pk=rf["pk"], q=None if q is None else q[t]))
    return out


def oracle_fv_step_nh(cs, gs, fl, dp_ref, st, ak, bk, bdt, k_split, remap_par, npz, q=None, last_step=False, fill2d=()):
    """nonhydrostatic k_split loop on six faces over the oracle: dyn_core substeps -> tracer_2d -> Lagrangian_to_Eulerian.
    With fl.use_cond / fl.moist_kappa the faces carry q_con / cappa (halo updates at fv_dynamics.F90:464-465 / :487-488)."""
    mdt = bdt / float(k_split)
    moist_names = (("q_con",) if fl.use_cond else ()) + (("cappa",) if fl.moist_kappa else ())
    cur = [{k: s[k].copy(order="F") for k in ("u", "v", "w", "delp", "pt", "delz", "phis") + moist_names} for s in st]
    q = None if q is None else [x.copy(order="F") for x in q]
    out = None
    bd = gs[0].bd
    for n_map in range(1, k_split + 1):
        dp1 = [c["delp"].copy(order="F") for c in cur]
        if moist_names:
            exchange(cs, cur, moist_names, "A")
        if fl.inline_q and q is not None:
            for t in range(6):
                cur[t]["q"] = q[t]
        f = oracle_substeps_nh(cs, gs, fl, dp_ref, cur, mdt, npz)
        if fl.inline_q and q is not None:
            q = [x["q"] for x in f]
        elif q is not None:
            _oracle_tracers(cs, gs, fl, npz, q, dp1, f)
            _oracle_fill2d(cs, gs, npz, q, [x["delp"] for x in f], fill2d if fl.hord_tr < 8 else ())
        out = []
        for t in range(6):
            x = f[t]
            rf = dict(ps=bd.zeros("A"), pe=x["pe"], delp=x["delp"], pkz=bd.zeros("CC", npz), pk=x["pk"], u=x["u"], v=x["v"], w=x["w"],
                      delz=x["delz"], pt=x["pt"], peln=x["peln"], omga=x["omga"], ws=x["ws"])
            if q is not None:
                rf["q"] = q[t]
            for n in moist_names:
                rf[n] = x[n]
            if remap_par.get("remap_te"):
                rf["hs"], rf["te"] = st[t]["phis"], bd.zeros("A", npz)
            O.lagrangian_to_eulerian(gs[t], npz, dict(remap_par, last_step=int(last_step and n_map == k_split)), rf, ak, bk)
            cur[t] = dict(u=rf["u"], v=rf["v"], w=rf["w"], delp=rf["delp"], pt=rf["pt"], delz=rf["delz"], phis=st[t]["phis"])
            cur[t].update({n: x[n] for n in ("du", "dv") if n in x})
            for n in moist_names:
                cur[t][n] = rf[n]
            out.append(dict(cur[t], pkz=rf["pkz"], ps=rf["ps"], pe=rf["pe"], peln=rf["peln"], pk=rf["pk"], q=None if q is None else q[t]))
    return out


def oracle_tracer_2d(cs, gs, npz, nq, q, dp1, mfx, mfy, cx, cy, hord, q_split=0, nord_tr=0, trdm=0.0):
    """tracer_2d (fv_tracer2d.F90:297-557) on six faces over the oracle's pieces: the Courant-number maximum reduced over the
    faces, the q halo updates between the sub-cycles.  q, dp1, mfx, ... are lists of six arrays, updated in place."""
    xfx = [np.zeros_like(c) for c in cx]
    yfx = [np.zeros_like(c) for c in cy]
    cmax = np.max(np.stack([O.tracer_2d_prep(gs[t], npz, q_split, cx[t], cy[t], xfx[t], yfx[t]) for t in range(6)]), axis=0)
    nsplt = int(1.0 + float(np.max(cmax))) if q_split == 0 else q_split
    if nsplt != 1:
        ksplt = (1.0 + cmax).astype(np.int32)
        frac = 1.0 / ksplt.astype(np.float64)
        for t in range(6):
            O.tracer_2d_scale(gs[t], npz, frac, cx[t], xfx[t], mfx[t], cy[t], yfx[t], mfy[t])
    else:
        ksplt = np.ones(npz, dtype=np.int32)
    if trdm > 1.0e-4:
        cs.topo.update("A", dp1)                     # dp1_pack, fv_tracer2d.F90:466
    for it in range(1, nsplt + 1):
        cs.topo.update("A", q)
        for t in range(6):
            O.tracer_2d_step(gs[t], npz, nq, it, nsplt, ksplt, q[t], dp1[t], mfx[t], mfy[t], cx[t], cy[t], xfx[t], yfx[t], hord,
                             nord_tr, trdm)
    return nsplt
