"""Parity checks of the nonhydrostatic column path (library vs oracle), shared by the GPU tests and
the host-emulation logic tests.  exp/log are evaluated by different math libraries on the host
(glibc) and the device (ROCm ocml), neither correctly rounded, so these kernels cannot be bit-exact
on the GPU: the tolerance is the one BASELINE.json states (relative RMS < 1e-12); the host-emulation
build shares glibc with the oracle and is checked at 1e-14."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
import parity_common as P
from fields import smooth_state
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill
from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, KAPPA, RDGAS, Context, nh_consts
from test_oracle_properties import default_levels

from gfdl_atmos_cubed_sphere_amd.synthetic import PTOP, nh_state  # noqa: E402,F401


def _tol(lib):
    return 1e-14


def check_update_dz_c(lib, nx=24, ny=13, km=6):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, True)
    s = nh_state(bd, km)
    st = smooth_state(bd, km)
    f = P.run_c_sw_oracle(g, bd, km, st, 3.0, False)
    gz = s["zh"].copy(order="F")
    ws = bd.zeros("A")
    O.update_dz_c(g, km, 3.0, s["dp0"], s["zs"], f["ut"], f["vt"], gz, ws)
    ctx = Context(g, km, lib=lib)
    try:
        ctx.set_dp_ref(s["dp0"])
        d_gz, d_ws = ctx.zeros("A", km + 1), ctx.zeros("A")
        ctx.update_dz_c(3.0, ctx.from_host(s["zs"]), ctx.from_host(f["ut"]), ctx.from_host(f["vt"]),
                        ctx.from_host(s["zh"]), d_gz, d_ws)
        r = (bd.is_ - 1, bd.ie + 1, bd.js - 1, bd.je + 1)
        P.assert_close("gz", bd.view(d_gz.download(), "A", *r), bd.view(gz, "A", *r), _tol(lib))
        P.assert_close("ws", bd.view(d_ws.download(), "A", *r), bd.view(ws, "A", *r), _tol(lib))
    finally:
        ctx.close()


def moist_fields(bd, km, seed=17):
    """q_con in [0, 0.02] and cappa around kappa: the use_cond / moist_kappa inputs of the Riemann solvers"""
    rng = np.random.default_rng(seed)
    q_con = np.asfortranarray(0.02 * rng.uniform(0, 1, bd.shape("A", km)))
    cappa = np.asfortranarray((2.0 / 7.0) * (1.0 - 0.1 * rng.uniform(0, 1, bd.shape("A", km))))
    return q_con, cappa


def _riem_context(g, km, lib, lds):
    """lds: the dry SIM1 solvers with the levels across the lanes (csrc/nh_fast.h RiemFast<CG, true>, the default where it is built;
    bit-identical to the slab kernels); False: FV3_MI355X_RIEM_LDS=0, the slab kernels (csrc/nh_kernels.h) for every configuration"""
    import os
    saved = os.environ.pop("FV3_MI355X_RIEM_LDS", None)
    if not lds:
        os.environ["FV3_MI355X_RIEM_LDS"] = "0"      # read when the context is created
    try:
        return Context(g, km, lib=lib)
    finally:
        os.environ.pop("FV3_MI355X_RIEM_LDS", None)
        if saved is not None:
            os.environ["FV3_MI355X_RIEM_LDS"] = saved


def tau_w_profile(km, dt_c, tau_w=25.0):
    """rff(1:k_rf) of fast_tau_w_sec = tau_w (nh_utils.F90:356-367) for a column whose upper half is above rf_cutoff"""
    pfull = PTOP * 1.2 + (1.0e5 - PTOP) * (np.arange(km) + 0.5) / km
    rff = O.fast_tau_w_rff(km, dt_c, tau_w, float(pfull[km // 2]) + 1.0, PTOP, pfull)
    assert 0 < len(rff) <= km and np.all(rff < 1.0) and np.all(rff > 0.0)
    return rff


def check_riem_solver_c(lib, nx=24, ny=13, km=8, a_imp=1.0, use_cond=False, moist_kappa=False, lds=True, out=None, m_split=1,
                        tau_w=0.0):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    s = nh_state(bd, km)
    cn = nh_consts(PTOP, a_imp=a_imp, m_split=m_split)
    rng = np.random.default_rng(3)
    ws = np.asfortranarray(0.1 * rng.uniform(-1, 1, bd.shape("A")))
    hs = np.asfortranarray(s["zs"] * GRAV)
    gz = s["zh"].copy(order="F")
    pef = bd.zeros("A", km + 1)
    q_con, cappa = moist_fields(bd, km)
    q_con, cappa = (q_con if use_cond else None), (cappa if moist_kappa else None)
    rff = tau_w_profile(km, 3.0, tau_w) if tau_w > 0.0 else None   # fast_tau_w_sec > 0: w2(k) * rff(k) inside SIM1_solver
    O.set_fast_tau_w(rff)
    try:
        O.riem_solver_c(g, km, 3.0, cn, hs, s["w"], s["pt"], s["delp"], gz, pef, ws, q_con, cappa)
    finally:
        O.set_fast_tau_w(None)
    if rff is not None:   # the damping really changes the answer
        gz0, pef0 = s["zh"].copy(order="F"), bd.zeros("A", km + 1)
        O.riem_solver_c(g, km, 3.0, cn, hs, s["w"], s["pt"], s["delp"], gz0, pef0, ws, q_con, cappa)
        assert P.rel_rms(pef0, pef) > 1e-10
    if use_cond:   # the moist branch really changes the answer
        gz0, pef0 = s["zh"].copy(order="F"), bd.zeros("A", km + 1)
        O.riem_solver_c(g, km, 3.0, cn, hs, s["w"], s["pt"], s["delp"], gz0, pef0, ws)
        assert P.rel_rms(pef0, pef) > 1e-8
    ctx = _riem_context(g, km, lib, lds)
    try:
        d_gz, d_pef = ctx.from_host(s["zh"]), ctx.zeros("A", km + 1)
        ctx.set_condensate(ctx.from_host(q_con) if use_cond else None, ctx.from_host(cappa) if moist_kappa else None)
        ctx.set_fast_tau_w(rff)
        ctx.riem_solver_c(3.0, cn, ctx.from_host(hs), ctx.from_host(s["w"]), ctx.from_host(s["pt"]),
                          ctx.from_host(s["delp"]), d_gz, d_pef, ctx.from_host(ws))
        r = (bd.is_ - 1, bd.ie + 1, bd.js - 1, bd.je + 1)
        tol = _tol(lib)
        e1 = P.assert_close("gz", bd.view(d_gz.download(), "A", *r), bd.view(gz, "A", *r), tol)
        e2 = P.assert_close("pef", bd.view(d_pef.download(), "A", *r), bd.view(pef, "A", *r), tol)
        if out is not None:
            out.update(gz=bd.view(d_gz.download(), "A", *r), pef=bd.view(d_pef.download(), "A", *r))
    finally:
        ctx.close()
    return max(e1, e2)


def check_riem_solver3(lib, nx=24, ny=13, km=8, a_imp=1.0, use_logp=False, last_call=True, fp_out=False, use_cond=False,
                       moist_kappa=False, lds=True, out=None, m_split=1, tau_w=0.0):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    s = nh_state(bd, km)
    cn = nh_consts(PTOP, a_imp=a_imp, m_split=m_split)
    rng = np.random.default_rng(4)
    ws = np.asfortranarray(0.1 * rng.uniform(-1, 1, bd.shape("CC")))
    q_con, cappa = moist_fields(bd, km)
    q_con, cappa = (q_con if use_cond else None), (cappa if moist_kappa else None)
    o = dict(w=s["w"].copy(order="F"), zh=s["zh"].copy(order="F"), delz=bd.zeros("CC", km),
             ppe=bd.zeros("A", km + 1), pk3=bd.full("A", 1e40, km + 1), pk=bd.zeros("CC", km + 1),
             pe=np.zeros((nx + 2, km + 1, ny + 2), order="F"), peln=np.zeros((nx, km + 1, ny), order="F"))
    rff = tau_w_profile(km, 3.0, tau_w) if tau_w > 0.0 else None   # the profile is Riem_Solver_c's (half the acoustic step)
    O.set_fast_tau_w(rff)
    try:
        O.riem_solver3(g, km, 6.0, cn, s["zs"], o["w"], o["delz"], s["pt"], s["delp"], o["zh"], o["pe"], o["ppe"],
                       o["pk3"], o["pk"], o["peln"], ws, use_logp, last_call, fp_out, q_con, cappa)
    finally:
        O.set_fast_tau_w(None)
    if rff is not None:   # the damping really changes the answer
        o0 = dict(w=s["w"].copy(order="F"), zh=s["zh"].copy(order="F"), delz=bd.zeros("CC", km),
                  ppe=bd.zeros("A", km + 1), pk3=bd.full("A", 1e40, km + 1), pk=bd.zeros("CC", km + 1),
                  pe=np.zeros((nx + 2, km + 1, ny + 2), order="F"), peln=np.zeros((nx, km + 1, ny), order="F"))
        O.riem_solver3(g, km, 6.0, cn, s["zs"], o0["w"], o0["delz"], s["pt"], s["delp"], o0["zh"], o0["pe"], o0["ppe"],
                       o0["pk3"], o0["pk"], o0["peln"], ws, use_logp, last_call, fp_out, q_con, cappa)
        assert P.rel_rms(bd.view(o0["w"], "A", bd.is_, bd.ie, bd.js, bd.je), bd.view(o["w"], "A", bd.is_, bd.ie, bd.js, bd.je)) > 1e-6
    if use_cond or moist_kappa:   # the moist branches really change the answer
        o0 = dict(w=s["w"].copy(order="F"), zh=s["zh"].copy(order="F"), delz=bd.zeros("CC", km),
                  ppe=bd.zeros("A", km + 1), pk3=bd.full("A", 1e40, km + 1), pk=bd.zeros("CC", km + 1),
                  pe=np.zeros((nx + 2, km + 1, ny + 2), order="F"), peln=np.zeros((nx, km + 1, ny), order="F"))
        O.riem_solver3(g, km, 6.0, cn, s["zs"], o0["w"], o0["delz"], s["pt"], s["delp"], o0["zh"], o0["pe"], o0["ppe"],
                       o0["pk3"], o0["pk"], o0["peln"], ws, use_logp, last_call, fp_out)
        assert P.rel_rms(o0["delz"], o["delz"]) > 1e-8
    ctx = _riem_context(g, km, lib, lds)
    worst = 0.0
    try:
        ctx.set_condensate(ctx.from_host(q_con) if use_cond else None, ctx.from_host(cappa) if moist_kappa else None)
        d = {k: ctx.from_host(v) for k, v in dict(w=s["w"], zh=s["zh"], delz=bd.zeros("CC", km),
                                                   ppe=bd.zeros("A", km + 1), pk3=bd.full("A", 1e40, km + 1),
                                                   pk=bd.zeros("CC", km + 1),
                                                   pe=np.zeros((nx + 2, km + 1, ny + 2), order="F"),
                                                   peln=np.zeros((nx, km + 1, ny), order="F")).items()}
        ctx.set_fast_tau_w(rff)
        ctx.riem_solver3(6.0, cn, ctx.from_host(s["zs"]), d["w"], d["delz"], ctx.from_host(s["pt"]),
                         ctx.from_host(s["delp"]), d["zh"], d["pe"], d["ppe"], d["pk3"], d["pk"], d["peln"],
                         ctx.from_host(ws), use_logp, last_call, fp_out)
        r = (bd.is_, bd.ie, bd.js, bd.je)
        tol = _tol(lib)
        for n in ("w", "zh", "ppe", "pk3"):
            worst = max(worst, P.assert_close(n, bd.view(d[n].download(), "A", *r), bd.view(o[n], "A", *r), tol))
        worst = max(worst, P.assert_close("delz", d["delz"].download(), o["delz"], tol))
        if last_call:
            worst = max(worst, P.assert_close("pk", d["pk"].download(), o["pk"], tol))
            worst = max(worst, P.assert_close("peln", d["peln"].download(), o["peln"], tol))
            worst = max(worst, P.assert_close("pe", d["pe"].download()[1:-1, :, 1:-1], o["pe"][1:-1, :, 1:-1], tol))
        if out is not None:
            out.update({n: bd.view(d[n].download(), "A", *r) for n in ("w", "zh", "ppe", "pk3")}, delz=d["delz"].download())
            if last_call:
                out.update(pk=d["pk"].download(), peln=d["peln"].download(), pe=d["pe"].download()[1:-1, :, 1:-1])
    finally:
        ctx.close()
    return worst


def check_update_dz_d(lib, nx=40, ny=19, km=5, lev_over=None, hord=10, lds=True, out=None):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(8)
    arr = {n: bd.zeros(k, km) for n, k in (("crx", "CX"), ("xfx", "CX"), ("cry", "CY"), ("yfx", "CY"))}
    for k in range(km):
        c = P._courant(bd, g, rng, cmax=0.4)
        for n, a in zip(("crx", "cry", "xfx", "yfx"), c[:4]):
            arr[n][:, :, k] = a
    lev = default_levels(km, **(lev_over or {}))
    ndif = np.concatenate([lev["nord_v"], lev["nord_v"][-1:]]).astype(np.int32)
    damp = np.concatenate([lev["damp_vt"], lev["damp_vt"][-1:]])
    zh = s["zh"].copy(order="F")
    ws = bd.zeros("CC")
    rdt = 1.0 / 6.0
    O.update_dz_d(g, km, ndif, damp, hord, s["dp0"], s["zs"], zh, arr["crx"], arr["cry"], arr["xfx"], arr["yfx"], ws, rdt)
    ctx = _riem_context(g, km, lib, lds)    # lds: edge_profile with the levels across the lanes (EdgeProfileLds); False: the slab kernel
    try:
        ctx.set_dp_ref(s["dp0"])
        ctx.dsw_levels(lev)
        d_out, d_ws = ctx.zeros("A", km + 1), ctx.zeros("CC")
        ctx.update_dz_d(hord, ctx.from_host(s["zs"]), ctx.from_host(s["zh"]), d_out, ctx.from_host(arr["crx"]),
                        ctx.from_host(arr["cry"]), ctx.from_host(arr["xfx"]), ctx.from_host(arr["yfx"]), d_ws, rdt)
        r = (bd.is_, bd.ie, bd.js, bd.je)
        e = P.assert_close("zh", bd.view(d_out.download(), "A", *r), bd.view(zh, "A", *r), _tol(lib))
        P.assert_close("ws", d_ws.download(), ws, _tol(lib))
        if out is not None:
            out.update(zh=bd.view(d_out.download(), "A", *r), ws=d_ws.download())
        return e
    finally:
        ctx.close()


def _pressure_fields(bd, km, s, rng):
    pe = PTOP + np.concatenate([np.zeros(bd.shape("A") + (1,)), np.cumsum(s["delp"], axis=2)], axis=2)
    pk = np.asfortranarray(pe ** KAPPA)
    pp = np.asfortranarray(50.0 * rng.uniform(-1, 1, pe.shape))
    pp[:, :, 0] = 0.0
    gz = np.asfortranarray(s["zh"] * GRAV)
    return np.asfortranarray(pe), pk, pp, gz


def check_p_grad_c(lib, nx=24, ny=13, km=5, hydrostatic=False):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(5)
    pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
    pkc = pk if hydrostatic else np.asfortranarray(pe + pp)
    st = smooth_state(bd, km)
    uc = np.asfortranarray(rng.uniform(-10, 10, bd.shape("V", km)))
    vc = np.asfortranarray(rng.uniform(-10, 10, bd.shape("U", km)))
    uc_o, vc_o = uc.copy(order="F"), vc.copy(order="F")
    O.p_grad_c(g, km, 3.0, s["delp"], pkc, gz, uc_o, vc_o, hydrostatic)
    ctx = Context(g, km, lib=lib)
    try:
        d_uc, d_vc = ctx.from_host(uc), ctx.from_host(vc)
        ctx.p_grad_c(3.0, ctx.from_host(s["delp"]), ctx.from_host(pkc), ctx.from_host(gz), d_uc, d_vc, hydrostatic)
        P.assert_close("uc", d_uc.download(), uc_o, _tol(lib))
        P.assert_close("vc", d_vc.download(), vc_o, _tol(lib))
    finally:
        ctx.close()


def check_nh_p_grad(lib, nx=40, ny=19, km=5, grid=None, out=None, fused=True):
    """fused (the default where the domain has no face edges): NhPGradFused, a2b_ord4 and the gradient in one kernel; False
    (FV3_MI355X_PGRAD_FUSED=0, read when the context is created): the corner values through memory"""
    import os
    saved = os.environ.pop("FV3_MI355X_PGRAD_FUSED", None)
    if not fused:
        os.environ["FV3_MI355X_PGRAD_FUSED"] = "0"
    try:
        return _check_nh_p_grad(lib, nx, ny, km, grid, out)
    finally:
        os.environ.pop("FV3_MI355X_PGRAD_FUSED", None)
        if saved is not None:
            os.environ["FV3_MI355X_PGRAD_FUSED"] = saved


def check_nh_p_grad_fused_bits(lib, **dims):
    """the one-kernel nh_p_grad against a2b_ord4 + the gradient: the same bits in u and v"""
    a, b = {}, {}
    check_nh_p_grad(lib, out=a, **dims)
    check_nh_p_grad(lib, out=b, fused=False, **dims)
    for n in a:
        assert np.array_equal(a[n], b[n]), f"nh_p_grad {dims}: {n} of the fused kernel differs from the two-kernel path"


def _check_nh_p_grad(lib, nx, ny, km, grid, out):
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(6)
    pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
    u = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("V", km)))
    top = PTOP ** KAPPA
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"), pp=pp.copy(order="F"), gz=gz.copy(order="F"),
             delp=s["delp"].copy(order="F"), pk=pk.copy(order="F"))
    O.nh_p_grad(g, km, o["u"], o["v"], o["pp"], o["gz"], o["delp"], o["pk"], 6.0, top)
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v = ctx.from_host(u), ctx.from_host(v)
        ctx.nh_p_grad(d_u, d_v, ctx.from_host(pp), ctx.from_host(gz), ctx.from_host(s["delp"]), ctx.from_host(pk), 6.0, top)
        P.assert_close("u", bd.view(d_u.download(), "U", bd.is_, bd.ie, bd.js, bd.je + 1),
                       bd.view(o["u"], "U", bd.is_, bd.ie, bd.js, bd.je + 1), _tol(lib))
        P.assert_close("v", bd.view(d_v.download(), "V", bd.is_, bd.ie + 1, bd.js, bd.je),
                       bd.view(o["v"], "V", bd.is_, bd.ie + 1, bd.js, bd.je), _tol(lib))
        if out is not None:
            out.update(u=d_u.download(), v=d_v.download())
    finally:
        ctx.close()


def check_split_p_grad(lib, nx=40, ny=19, km=5, beta=0.4, grid=None):
    """split_p_grad (dyn_core.F90:1795-1900) over two calls: beta_d = 0 with du = dv = 0, then beta with the saved gradient"""
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(26)
    u = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("V", km)))
    top = PTOP ** KAPPA
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"), du=bd.zeros("U", km), dv=bd.zeros("V", km))
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v, d_du, d_dv = ctx.from_host(u), ctx.from_host(v), ctx.zeros("U", km), ctx.zeros("V", km)
        for call, bt in enumerate((0.0, beta, beta)):
            pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
            O.split_p_grad(g, km, o["u"], o["v"], pp.copy(order="F"), gz.copy(order="F"), s["delp"].copy(order="F"),
                           pk.copy(order="F"), bt, 6.0, top, o["du"], o["dv"])
            ctx.split_p_grad(d_u, d_v, ctx.from_host(pp), ctx.from_host(gz), ctx.from_host(s["delp"]), ctx.from_host(pk), bt, 6.0, top,
                             d_du, d_dv)
            for n, kind, da, r in (("u", "U", d_u, (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", d_v, (bd.is_, bd.ie + 1, bd.js, bd.je)),
                                   ("du", "U", d_du, (bd.is_, bd.ie, bd.js, bd.je + 1)), ("dv", "V", d_dv, (bd.is_, bd.ie + 1, bd.js, bd.je))):
                P.assert_close(f"{n} call {call}", bd.view(da.download(), kind, *r), bd.view(o[n], kind, *r), _tol(lib))
    finally:
        ctx.close()


def check_grad1_p_update(lib, nx=40, ny=19, km=5, beta=0.4, d_ext=0.02, grid=None):
    """grad1_p_update (dyn_core.F90:2033-2116) over three calls, with and without the external-mode damping field"""
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(27)
    u = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("V", km)))
    vt = np.asfortranarray(rng.uniform(-1e-5, 1e-5, bd.shape("A", km)))
    top = PTOP ** KAPPA
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"), du=bd.zeros("U", km), dv=bd.zeros("V", km))
    divg2 = bd.zeros("A")
    O.divg2_ext(g, km, d_ext, s["delp"], vt, divg2)
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v, d_du, d_dv = ctx.from_host(u), ctx.from_host(v), ctx.zeros("U", km), ctx.zeros("V", km)
        d_d2 = ctx.from_host(divg2)
        for call, bt in enumerate((0.0, beta, beta)):
            pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
            O.grad1_p_update(g, km, divg2, o["u"], o["v"], pk.copy(order="F"), gz.copy(order="F"), 6.0, top, bt, o["du"], o["dv"])
            ctx.grad1_p_update(d_d2 if d_ext > 0 else None, d_u, d_v, ctx.from_host(pk), ctx.from_host(gz), 6.0, top, bt, d_du, d_dv)
            for n, kind, da, r in (("u", "U", d_u, (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", d_v, (bd.is_, bd.ie + 1, bd.js, bd.je)),
                                   ("du", "U", d_du, (bd.is_, bd.ie, bd.js, bd.je + 1)), ("dv", "V", d_dv, (bd.is_, bd.ie + 1, bd.js, bd.je))):
                P.assert_close(f"{n} call {call}", bd.view(da.download(), kind, *r), bd.view(o[n], kind, *r), _tol(lib))
    finally:
        ctx.close()


def check_one_grad_p_nh(lib, nx=40, ny=19, km=5, d_ext=0.02, grid=None):
    """one_grad_p with hydrostatic = .false. (dyn_core.F90:1909-2030, the beta < -0.1 call of the nonhydrostatic loop): pk = a full
    pressure, the layer weights a2b_ord4 of delp, gz = zh * grav formed inside"""
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(31)
    u = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("V", km)))
    vt = np.asfortranarray(rng.uniform(-1e-5, 1e-5, bd.shape("A", km)))
    divg2 = bd.zeros("A")
    O.divg2_ext(g, km, d_ext, s["delp"], vt, divg2)
    pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
    full = np.asfortranarray(pe + pp)                  # what Riem_Solver3 leaves in pkc with fp_out
    zh = np.asfortranarray(s["zh"])
    gzs = np.asfortranarray(zh * GRAV)
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"))
    O.one_grad_p_nh(g, km, 6.0, PTOP, divg2, o["u"], o["v"], full.copy(order="F"), gzs.copy(order="F"), s["delp"])
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v = ctx.from_host(u), ctx.from_host(v)
        ctx.one_grad_p_nh(d_u, d_v, ctx.from_host(full), ctx.from_host(zh), ctx.from_host(divg2) if d_ext > 0 else None,
                          ctx.from_host(s["delp"]), 6.0, PTOP, gz_scale=GRAV)
        for n, kind, da, r in (("u", "U", d_u, (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", d_v, (bd.is_, bd.ie + 1, bd.js, bd.je))):
            P.assert_close(f"one_grad_p_nh {n}", bd.view(da.download(), kind, *r), bd.view(o[n], kind, *r), _tol(lib))
    finally:
        ctx.close()


def check_halos_and_geopk(lib, nx=24, ny=13, km=6):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    s = nh_state(bd, km)
    tol = _tol(lib)
    ctx = Context(g, km, lib=lib)
    try:
        d_delp = ctx.from_host(s["delp"])
        for use_logp in (False, True):
            pk3 = bd.full("A", 7.0, km + 1)
            ref = pk3.copy(order="F")
            O.pk3_halo(g, km, PTOP, KAPPA, ref, s["delp"], use_logp)
            d = ctx.from_host(pk3)
            ctx.pk3_halo(PTOP, KAPPA, d, d_delp, use_logp)
            P.assert_close("pk3_halo", d.download(), ref, tol)
        pe = np.full((nx + 2, km + 1, ny + 2), 3.0, order="F")
        ref = pe.copy(order="F")
        O.pe_halo(g, km, PTOP, ref, s["delp"])
        d = ctx.from_host(pe)
        ctx.pe_halo(PTOP, d, d_delp)
        got = d.download()
        # the four corner columns of the ring are not written by the reference either way they agree
        P.assert_close("pe_halo", got, ref, tol)
        for CG in (True, False):
            o = dict(pe=np.zeros((nx + 2, km + 1, ny + 2), order="F"), peln=np.zeros((nx, km + 1, ny), order="F"),
                     pk=bd.zeros("A", km + 1), gz=bd.zeros("A", km + 1), pkz=bd.zeros("CC", km))
            hs = np.asfortranarray(s["zs"] * GRAV)
            O.geopk(g, km, PTOP, KAPPA, CP_AIR, o["pe"], o["peln"], s["delp"], o["pk"], o["gz"], hs, s["pt"], o["pkz"], CG)
            dd = {k: ctx.from_host(np.zeros_like(v)) for k, v in o.items()}
            ctx.geopk(PTOP, KAPPA, CP_AIR, dd["pe"], dd["peln"], d_delp, dd["pk"], dd["gz"], ctx.from_host(hs),
                      ctx.from_host(s["pt"]), dd["pkz"], CG)
            for n in ("pk", "gz", "pe", "peln") + (() if CG else ("pkz",)):
                P.assert_close(f"geopk {n} CG={CG}", dd[n].download(), o[n], tol)
    finally:
        ctx.close()


def check_heat_source_path(lib, nx=24, ny=13, km=6, hydrostatic=False, n_con=None, nmax=2):
    """heat_source accumulation, del2_cubed and the heating application against the oracle"""
    from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, True)
    s = nh_state(bd, km, seed=5)
    rng = np.random.default_rng(8)
    n_con = km if n_con is None else n_con
    hs3 = np.asfortranarray(rng.uniform(-1, 1, bd.shape("A", km)) * 1.0e3)
    hs2 = np.asfortranarray(rng.uniform(-1, 1, bd.shape("CC", km)) * 1.0e3)
    delz = np.asfortranarray(np.diff(s["zh"], axis=2)[bd.ng:bd.ng + nx, bd.ng:bd.ng + ny, :])
    pkz = np.asfortranarray(rng.uniform(0.9, 1.1, bd.shape("CC", km)))
    pt = s["pt"].copy(order="F")
    # oracle
    r_hs = hs3.copy(order="F")
    r_hs[bd.ng:bd.ng + nx, bd.ng:bd.ng + ny, :] += hs2
    for k in range(km):
        periodic_fill(bd, r_hs[:, :, k], "A")
    filled = r_hs.copy(order="F")
    O.del2_cubed(g, km, 0.20 * g.da_min, nmax, r_hs)
    r_pt, r_pkz = pt.copy(order="F"), pkz.copy(order="F")
    O.apply_heat_source(g, km, n_con, hydrostatic, 37.5, 1.0, CP_AIR, CP_AIR - RDGAS, RDGAS, GRAV, r_pt, r_hs, s["delp"], delz,
                        r_pkz)
    ctx = Context(g, km, lib=lib)
    try:
        d_hs, d_hs2 = ctx.from_host(hs3), ctx.from_host(hs2)
        ctx.heat_source_accum(d_hs, d_hs2)
        got = d_hs.download()
        r = (bd.is_, bd.ie, bd.js, bd.je)
        P.assert_close("accum", bd.view(got, "A", *r), bd.view(filled, "A", *r), 0.0 + 1e-300)
        d_hs.upload(filled)
        ctx.del2_cubed(d_hs, 0.20 * g.da_min, nmax)
        d_pt, d_pkz = ctx.from_host(pt), ctx.from_host(pkz)
        ctx.apply_heat_source(n_con, hydrostatic, 37.5, 1.0, CP_AIR, CP_AIR - RDGAS, RDGAS, GRAV, d_pt, d_hs,
                              ctx.from_host(s["delp"]), ctx.from_host(delz), d_pkz)
        tol = 1e-14
        worst = P.assert_close("heat_source", bd.view(d_hs.download(), "A", *r), bd.view(r_hs, "A", *r), tol)
        worst = max(worst, P.assert_close("pt", bd.view(d_pt.download(), "A", *r), bd.view(r_pt, "A", *r), tol))
        worst = max(worst, P.assert_close("pkz", d_pkz.download(), r_pkz, tol))
    finally:
        ctx.close()
    return worst


def check_one_grad_p(lib, nx=40, ny=19, km=5, d_ext=0.02, grid=None):
    """external-mode divergence field + hydrostatic one_grad_p against the oracle"""
    bd = grid.bd if grid is not None else Bounds(1, nx, 1, ny)
    g = grid if grid is not None else P.make_grid(bd, True)
    s = nh_state(bd, km)
    rng = np.random.default_rng(16)
    pe, pk, pp, gz = _pressure_fields(bd, km, s, rng)
    u = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-1e4, 1e4, bd.shape("V", km)))
    vt = np.asfortranarray(rng.uniform(-1e-5, 1e-5, bd.shape("A", km)))
    top = PTOP ** KAPPA
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"), pk=pk.copy(order="F"), gz=gz.copy(order="F"))
    divg2 = bd.zeros("A")
    O.divg2_ext(g, km, d_ext, s["delp"], vt, divg2)
    O.one_grad_p_hydro(g, km, 6.0, top, divg2, o["u"], o["v"], o["pk"], o["gz"])
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v, d_d2 = ctx.from_host(u), ctx.from_host(v), ctx.zeros("A")
        ctx.divg2_ext(d_ext, ctx.from_host(s["delp"]), ctx.from_host(vt), d_d2)
        r = (bd.is_, bd.ie + 1, bd.js, bd.je + 1)
        if d_ext > 0:
            P.assert_close("divg2", bd.view(d_d2.download(), "A", *r), bd.view(divg2, "A", *r), _tol(lib))
        ctx.one_grad_p(d_u, d_v, ctx.from_host(pk), ctx.from_host(gz), d_d2 if d_ext > 0 else None, 6.0, top)
        P.assert_close("u", bd.view(d_u.download(), "U", bd.is_, bd.ie, bd.js, bd.je + 1),
                       bd.view(o["u"], "U", bd.is_, bd.ie, bd.js, bd.je + 1), _tol(lib))
        P.assert_close("v", bd.view(d_v.download(), "V", bd.is_, bd.ie + 1, bd.js, bd.je),
                       bd.view(o["v"], "V", bd.is_, bd.ie + 1, bd.js, bd.je), _tol(lib))
    finally:
        ctx.close()


def check_c2l_and_rayleigh(lib, nx=70, ny=33, km=12, hydrostatic=False, conserve=True, tau=0.02, rf_cutoff=30.e2):
    """cubed_to_latlon (c2l_ord 2 and 4) and Rayleigh_Friction (grid_type = 4 branches) against the oracle"""
    from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
    from fields import smooth_state
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st = smooth_state(bd, km, hydrostatic=hydrostatic, noise=0.05)
    rng = np.random.default_rng(21)
    ptop = 100.0
    pm = ptop * np.exp(np.linspace(0.1, 5.0, km))          # layer-mean pressures, increasing downwards
    dt = 225.0
    rf, kmax = O.rayleigh_rf(km, dt, tau, rf_cutoff, ptop, pm)
    assert 0 < kmax < km
    u, v, pt = st["u"].copy(order="F"), st["v"].copy(order="F"), st["pt"].copy(order="F")
    w = None if hydrostatic else np.asfortranarray(st["w"] * 10.0)
    for a, kind in ((u, "U"), (v, "V")):
        a *= 6.0                                              # strong winds: a damping of a few per cent
        for k in range(km):
            periodic_fill(bd, a[:, :, k], kind)
    delz = None if hydrostatic else np.asfortranarray(-rng.uniform(200., 400., bd.shape("CC", km)))
    r = (bd.is_, bd.ie, bd.js, bd.je)
    tol = 1e-14
    # ---- oracle ----
    ref = {}
    for o in (2, 4):
        ua, va = bd.zeros("A", km), bd.zeros("A", km)
        O.c2l(g, km, o, u, v, ua, va)
        ref[o] = (ua, va)
    r_u, r_v, r_pt = u.copy(order="F"), v.copy(order="F"), pt.copy(order="F")
    r_w = None if hydrostatic else w.copy(order="F")
    r_dz = None if hydrostatic else delz.copy(order="F")
    r_ua, r_va, r_u2f = bd.zeros("A", km), bd.zeros("A", km), bd.zeros("A", kmax)
    O.rayleigh_u2f(g, kmax, hydrostatic, r_u, r_v, r_w, r_ua, r_va, r_u2f)
    for k in range(kmax):
        periodic_fill(bd, r_u2f[:, :, k], "A")               # mpp_update_domains(u2f), fv_dynamics.F90:1208
    u2f_in = r_u2f.copy(order="F")
    O.rayleigh_apply(g, kmax, conserve, hydrostatic, CP_AIR, RDGAS, ptop, pm, rf, r_u2f, r_pt, r_dz, r_u, r_v, r_w)
    assert np.max(np.abs(r_u[:, :, 0] - u[:, :, 0])) > 1e-3 * np.max(np.abs(u[:, :, 0]))  # the test does something
    # ---- library ----
    ctx = Context(g, km, lib=lib)
    worst = 0.0
    try:
        d_u, d_v = ctx.from_host(u), ctx.from_host(v)
        for o in (2, 4):
            d_ua, d_va = ctx.zeros("A", km), ctx.zeros("A", km)
            ctx.c2l(o, d_u, d_v, d_ua, d_va)
            worst = max(worst, P.assert_close(f"ua{o}", bd.view(d_ua.download(), "A", *r), bd.view(ref[o][0], "A", *r), tol))
            worst = max(worst, P.assert_close(f"va{o}", bd.view(d_va.download(), "A", *r), bd.view(ref[o][1], "A", *r), tol))
        d_pt = ctx.from_host(pt)
        d_w = None if hydrostatic else ctx.from_host(w)
        d_dz = None if hydrostatic else ctx.from_host(delz)
        d_ua, d_va, d_u2f = ctx.zeros("A", km), ctx.zeros("A", km), ctx.zeros("A", km)
        ctx.rayleigh_u2f(kmax, hydrostatic, d_u, d_v, d_w, d_ua, d_va, d_u2f)
        got = d_u2f.download()
        worst = max(worst, P.assert_close("u2f", bd.view(got[:, :, :kmax], "A", *r), bd.view(u2f_in, "A", *r), tol))
        ctx.halo_fill_periodic(d_u2f, "A")
        ctx.rayleigh_apply(kmax, conserve, hydrostatic, CP_AIR, RDGAS, ptop, pm[:kmax], rf[:kmax], d_u2f, d_pt, d_dz, d_u,
                           d_v, d_w)
        worst = max(worst, P.assert_close("u", bd.view(d_u.download(), "U", bd.is_, bd.ie, bd.js, bd.je + 1),
                                          bd.view(r_u, "U", bd.is_, bd.ie, bd.js, bd.je + 1), tol))
        worst = max(worst, P.assert_close("v", bd.view(d_v.download(), "V", bd.is_, bd.ie + 1, bd.js, bd.je),
                                          bd.view(r_v, "V", bd.is_, bd.ie + 1, bd.js, bd.je), tol))
        worst = max(worst, P.assert_close("pt", bd.view(d_pt.download(), "A", *r), bd.view(r_pt, "A", *r), tol))
        if not hydrostatic:
            worst = max(worst, P.assert_close("w", bd.view(d_w.download(), "A", *r), bd.view(r_w, "A", *r), tol))
            worst = max(worst, P.assert_close("delz", d_dz.download(), r_dz, tol))
    finally:
        ctx.close()
    return worst


def check_mix_dp(lib, nx=37, ny=19, km=12, hydrostatic=False):
    """mix_dp (dyn_core.F90:2119-2200): a column state with layers far below 1 % of their reference thickness -- isolated ones, two in a row
    (the second is tested after it gave mass to the first), the top, the bottom layer, a NaN -- bit for bit against the oracle"""
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    rng = np.random.default_rng(17)
    sig = np.linspace(0.0, 1.0, km + 1) ** 1.5
    ak, bk = PTOP * (1.0 - sig), sig.copy()
    dref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    delp = np.asfortranarray(dref[None, None, :] * rng.uniform(0.8, 1.2, bd.shape("A", km)))
    thin = rng.uniform(0, 1, delp.shape) > 0.9
    delp[thin] *= 1.0e-3
    ng = bd.ng
    delp[ng + 3, ng + 2, 4:6] = dref[4:6] * 1.0e-4            # two in a row
    delp[ng + 5, ng + 4, 0] = 0.0                             # the top layer, empty
    delp[ng + 6, ng + 1, km - 1] = dref[km - 1] * 1.0e-5      # the bottom layer
    delp[ng + 7, ng + 7, 3] = np.nan                          # `.not. delp >= dpmin` catches NaN
    pt = np.asfortranarray(300.0 + 30.0 * rng.uniform(-1, 1, delp.shape))
    w = np.asfortranarray(rng.uniform(-2, 2, delp.shape))
    o = dict(delp=delp.copy(order="F"), pt=pt.copy(order="F"), w=w.copy(order="F"))
    O.mix_dp(g, km, hydrostatic, ak, bk, None if hydrostatic else o["w"], o["delp"], o["pt"])
    r = (bd.is_, bd.ie, bd.js, bd.je)
    assert np.sum(bd.view(o["delp"], "A", *r) != bd.view(delp, "A", *r)) > 50       # the fix acted
    ctx = Context(g, km, lib=lib)
    try:
        ctx.set_ak_bk(ak, bk)
        d_dp, d_pt, d_w = ctx.from_host(delp), ctx.from_host(pt), ctx.from_host(w)
        ctx.mix_dp(hydrostatic, None if hydrostatic else d_w, d_dp, d_pt)
        out = {}
        for n, dv in (("delp", d_dp), ("pt", d_pt)) + (() if hydrostatic else (("w", d_w),)):
            a, b = bd.view(dv.download(), "A", *r), bd.view(o[n], "A", *r)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            m = ~np.isnan(b)
            assert np.array_equal(a[m], b[m]), n
            out[n] = 0.0
        halo_same = dv.download()[:ng, :, :]
        assert np.array_equal(np.isnan(d_dp.download()[:ng]), np.isnan(delp[:ng]))     # only the compute domain is touched
    finally:
        ctx.close()
    return out


def check_ray_fast(lib, nx=37, ny=19, km=12, hydrostatic=False, tau=0.5, rf_cutoff=None, ks=None):
    """Ray_fast (dyn_core.F90:2485-2601): the profile of its first call and the damping with the momentum handed back, against the
    oracle.  rf_cutoff: default = between levels km/2 and km/2 + 1; ks: the call site's (levels of pure pressure)"""
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    rng = np.random.default_rng(23)
    pfull = PTOP * 1.2 + (1.0e5 - PTOP) * (np.arange(km) + 0.5) / km
    dp = rng.uniform(500.0, 1500.0, km)
    rf_cutoff = float(pfull[km // 2]) + 1.0 if rf_cutoff is None else rf_cutoff
    ks = km - 2 if ks is None else ks
    kmax, k_rf, dm, rf = O.ray_fast_profile(km, ks, 12.0, tau, rf_cutoff, PTOP, pfull, dp)
    u = np.asfortranarray(rng.uniform(-30, 30, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-30, 30, bd.shape("V", km)))
    w = np.asfortranarray(rng.uniform(-2, 2, bd.shape("A", km)))
    o = dict(u=u.copy(order="F"), v=v.copy(order="F"), w=w.copy(order="F"))
    O.ray_fast(g, km, kmax, k_rf, rf, dp, hydrostatic, o["u"], o["v"], None if hydrostatic else o["w"])
    assert P.rel_rms(o["u"], u) > 1e-6 or not np.any(rf < 1.0)
    ctx = Context(g, km, lib=lib)
    try:
        d_u, d_v, d_w = ctx.from_host(u), ctx.from_host(v), ctx.from_host(w)
        ctx.set_ray_fast(kmax, k_rf, dm if k_rf > 0 else 1.0, rf[:kmax], dp)
        ctx.ray_fast(d_u, d_v, None if hydrostatic else d_w, hydrostatic)
        # the reference's statements on the same operands: bit for bit (the halo rows stay as they were)
        assert np.array_equal(d_u.download(), o["u"]) and np.array_equal(d_v.download(), o["v"]) and np.array_equal(d_w.download(), o["w"])
        # what the levels k <= kmax lose comes back on the levels k <= k_rf: the column's momentum sum dp u is conserved to rounding
        r = (bd.is_, bd.ie, bd.js, bd.je + 1)
        m0 = np.einsum("ijk,k->ij", bd.view(u, "U", *r), dp)
        m1 = np.einsum("ijk,k->ij", bd.view(d_u.download(), "U", *r), dp)
        if k_rf > 0:
            assert np.max(np.abs(m1 - m0)) <= 1e-11 * np.max(np.abs(m0))
    finally:
        ctx.close()
    return kmax, k_rf


def check_consv_am_kernels(lib, nx=37, ny=19, km=9):
    """compute_aam (fv_dynamics.F90:1266-1314) and the consv_am wind correction (:784-798): the reference's statements, bit for bit"""
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    rng = np.random.default_rng(29)
    coslat = np.asfortranarray(rng.uniform(0.05, 1.0, bd.shape("A")))
    ua = np.asfortranarray(rng.uniform(-60, 60, bd.shape("A", km)))
    delp = np.asfortranarray(rng.uniform(300, 1500, bd.shape("A", km)))
    u = np.asfortranarray(rng.uniform(-30, 30, bd.shape("U", km)))
    v = np.asfortranarray(rng.uniform(-30, 30, bd.shape("V", km)))
    lu, lv = np.asfortranarray(rng.uniform(-1, 1, bd.shape("U"))), np.asfortranarray(rng.uniform(-1, 1, bd.shape("V")))
    aam, mf, ps = bd.zeros("CC"), bd.zeros("CC"), bd.zeros("A")
    O.compute_aam(g, km, 6.3712e6, 7.292e-5, 1.0 / GRAV, PTOP, coslat, ua, delp, aam, mf, ps)
    ou, ov = u.copy(order="F"), v.copy(order="F")
    O.consv_am_apply(g, km, 0.37, lu, lv, ou, ov)
    ctx = Context(g, km, lib=lib)
    try:
        d_aam, d_mf, d_ps, d_u, d_v = ctx.zeros("CC"), ctx.zeros("CC"), ctx.zeros("A"), ctx.from_host(u), ctx.from_host(v)
        ctx.compute_aam(6.3712e6, 7.292e-5, 1.0 / GRAV, PTOP, ctx.from_host(coslat), ctx.from_host(ua), ctx.from_host(delp), d_aam, d_mf, d_ps)
        ctx.consv_am_apply(0.37, ctx.from_host(lu), ctx.from_host(lv), d_u, d_v)
        assert np.array_equal(d_aam.download(), aam) and np.array_equal(d_mf.download(), mf) and np.array_equal(d_ps.download(), ps)
        assert np.array_equal(d_u.download(), ou) and np.array_equal(d_v.download(), ov)
        assert np.all(aam != 0.0) and P.rel_rms(ou, u) > 1e-4
    finally:
        ctx.close()


def np_moist_cv(q, mp, cv_air):
    """moist_cv (fv_thermodynamics.F90:250-325) on whole arrays; q: (.., nq); returns cvm, q_con"""
    Q = lambda n: q[..., n - 1] if n > 0 else 0.0
    qv = Q(mp["sphum"])
    nwat = mp["nwat"]
    if nwat == 6:
        ql = Q(mp["liq_wat"]) + Q(mp["rainwat"])
        qs = Q(mp["ice_wat"]) + Q(mp["snowwat"]) + Q(mp["graupel"])
    elif nwat == 3:
        ql, qs = Q(mp["liq_wat"]), Q(mp["ice_wat"])
    else:
        raise NotImplementedError(nwat)
    q_con = ql + qs
    cvm = (1.0 - (qv + q_con)) * cv_air + qv * mp["cv_vap"] + ql * mp["c_liq"] + qs * mp["c_ice"]
    return cvm, q_con


def check_pt_to_theta_v(lib, nx=30, ny=17, km=6, hydrostatic=False, moist_kappa=False, use_cond=False, with_qv=True,
                        split=False):
    """fv3_pt_to_theta_v (fv_dynamics.F90:296-329, :379-399) against the same expressions in numpy; split = the
    pkz-only call followed by the conversion with that pkz (the order around Rayleigh_Friction)"""
    import parity_remap as R
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    s = nh_state(bd, km, seed=9)
    rng = np.random.default_rng(12)
    nq = 7
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", km) + (nq,)))
    q[..., 0] *= 0.02
    q[..., 1:6] *= 0.002
    mp = dict(R.MOIST6, sphum=1, moist_kappa=int(moist_kappa), use_cond=int(use_cond))
    zvir = 0.6077 if with_qv else 0.0
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + ny))
    delz = np.asfortranarray(np.diff(s["zh"], axis=2)[c])
    T = s["pt"].copy(order="F")
    T[c] = 250.0 + 30.0 * rng.uniform(0, 1, (nx, ny, km))
    pkz_in = np.asfortranarray(rng.uniform(0.9, 1.1, bd.shape("CC", km)))
    q_con0 = np.asfortranarray(0.01 * rng.uniform(0, 1, bd.shape("A", km)))
    # ---- numpy reference ----
    rdg = -RDGAS / GRAV
    dp1 = zvir * q[c + (slice(None), 0)] if with_qv else 0.0
    Tc, dpc = T[c], s["delp"][c]
    q_con_ref, cappa_ref = q_con0.copy(order="F"), bd.zeros("A", km)
    if hydrostatic:
        pkz_ref = pkz_in.copy(order="F")
    elif moist_kappa:
        cvm, qc = np_moist_cv(q[c], mp, CP_AIR - RDGAS)
        cap = RDGAS / (RDGAS + cvm / (1.0 + dp1))
        q_con_ref[c], cappa_ref[c] = qc, cap
        pkz_ref = O.fexp(cap * O.flog(rdg * dpc * Tc * (1.0 + dp1) * (1.0 - qc) / delz))
    else:
        pkz_ref = O.fexp((2.0 / 7.0) * O.flog(rdg * dpc * Tc * (1.0 + dp1) / delz))
    th_ref = T.copy(order="F")
    th_ref[c] = Tc * (1.0 + dp1) * (1.0 - q_con_ref[c]) / pkz_ref if use_cond else Tc * (1.0 + dp1) / pkz_ref
    # ---- library ----
    ctx = Context(g, km, lib=lib)
    try:
        d_q, d_qcon, d_cappa = ctx.from_host(q), ctx.from_host(q_con0), ctx.zeros("A", km)
        if moist_kappa or use_cond:
            ctx.set_moist(mp, d_qcon, d_cappa)
        d_pt, d_pkz = ctx.from_host(T), ctx.from_host(pkz_in)
        args = (zvir, 2.0 / 7.0, RDGAS, GRAV, d_pt, ctx.from_host(s["delp"]), None if hydrostatic else ctx.from_host(delz),
                d_q if with_qv else None, d_pkz)
        if split and not hydrostatic:
            ctx.pt_to_theta_v(-1, *args)
            assert np.array_equal(d_pt.download(), T)          # pkz only: pt untouched
            ctx.pt_to_theta_v(1, *args)
        else:
            ctx.pt_to_theta_v(int(hydrostatic), *args)
        tol = 1e-13
        worst = P.assert_close("pt", bd.view(d_pt.download(), "A", *r), bd.view(th_ref, "A", *r), tol)
        worst = max(worst, P.assert_close("pkz", d_pkz.download(), pkz_ref, tol))
        if moist_kappa and not hydrostatic:
            worst = max(worst, P.assert_close("q_con", bd.view(d_qcon.download(), "A", *r), bd.view(q_con_ref, "A", *r), tol))
            worst = max(worst, P.assert_close("cappa", bd.view(d_cappa.download(), "A", *r), bd.view(cappa_ref, "A", *r), tol))
    finally:
        ctx.close()
    return worst


def check_riem_lds_bits(lib, **dims):
    """the Riemann solvers with the levels across the lanes and the recurrences in the reference's order (nh_fast.h RiemFast<CG, true>,
    the default) against the slab kernels (FV3_MI355X_RIEM_LDS=0): the SAME BITS in every output, both within the parity tolerance of
    the oracle"""
    for kw in (dict(), dict(use_logp=True, last_call=True, fp_out=True), dict(last_call=False), dict(a_imp=0.75),
               dict(a_imp=0.75, use_logp=True, last_call=True, fp_out=True),   # a_imp < 1: SIM_solver (RiemFast<false, true, true>)
               dict(tau_w=25.0), dict(a_imp=0.75, tau_w=25.0)):                # fast_tau_w_sec > 0
        a, b = {}, {}
        check_riem_solver3(lib, out=a, **kw, **dims)
        check_riem_solver3(lib, lds=False, out=b, **kw, **dims)
        for n in a:
            assert np.array_equal(a[n], b[n]), f"riem_solver3 {kw}: {n} differs from the slab kernel"
    for kw in (dict(), dict(tau_w=25.0)):
        a, b = {}, {}
        check_riem_solver_c(lib, out=a, **kw, **dims)
        check_riem_solver_c(lib, lds=False, out=b, **kw, **dims)
        for n in a:
            assert np.array_equal(a[n], b[n]), f"riem_solver_c {kw}: {n} differs from the slab kernel"
    # use_cond / moist_kappa (RiemFast<CG, true, SIM, true>: nh_core.F90:96-166, nh_utils.F90:383-438)
    for mk in (dict(use_cond=True), dict(moist_kappa=True), dict(use_cond=True, moist_kappa=True)):
        for kw in (dict(use_logp=True, last_call=True, fp_out=True), dict(a_imp=0.75)):
            a, b = {}, {}
            check_riem_solver3(lib, out=a, **kw, **mk, **dims)
            check_riem_solver3(lib, lds=False, out=b, **kw, **mk, **dims)
            for n in a:
                assert np.array_equal(a[n], b[n]), f"riem_solver3 {mk} {kw}: {n} differs from the slab kernel"
        a, b = {}, {}
        check_riem_solver_c(lib, out=a, **mk, **dims)
        check_riem_solver_c(lib, lds=False, out=b, **mk, **dims)
        for n in a:
            assert np.array_equal(a[n], b[n]), f"riem_solver_c {mk}: {n} differs from the slab kernel"


def check_edge_profile_lds_bits(lib, **kw):
    """update_dz_d with edge_profile's levels across the lanes and its elimination in the reference's order (nh_fast.h EdgeProfileLds,
    the default) against the slab kernel (FV3_MI355X_RIEM_LDS=0): the same bits in zh and ws"""
    a, b = {}, {}
    check_update_dz_d(lib, out=a, **kw)
    check_update_dz_d(lib, lds=False, out=b, **kw)
    for n in a:
        assert np.array_equal(a[n], b[n]), f"update_dz_d {kw}: {n} differs from the slab kernel's"
