"""Whole-substep parity: the product's DynCore (device kernels + halo kernels + ping-pong) against
the oracle-orchestrated substep loop on the same nearly hydrostatic state."""
from __future__ import annotations

import numpy as np

import oracle_dyn_core as OD
import oracle_lib as O
import parity_common as P
import parity_nh as N
from fields import smooth_state
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynCore, DynFlags
from gfdl_atmos_cubed_sphere_amd.layout import Bounds
from gfdl_atmos_cubed_sphere_amd.lib import GRAV, Context


from gfdl_atmos_cubed_sphere_amd.synthetic import balanced_nh_state as make_state  # noqa: E402


def ulp_sensitivity(g, npz, fl, dp0, st, bdt, ref):
    """rel-RMS change of the ORACLE's own result when pt is perturbed by +-1 ulp: the conditioning floor of
    each output.  w is the sensitive one: the nonhydrostatic pressure perturbation is the difference of two
    O(1e5 Pa) numbers obtained through exp/log (nh_utils.F90:1299), so one ulp in them is O(1e-11) in w --
    the reference itself warns that this routine is not reproducible across transcendental implementations
    (nh_utils.F90:483-484)."""
    rng = np.random.default_rng(0)
    st2 = {k: v.copy() for k, v in st.items()}
    st2["pt"] = np.asfortranarray(st["pt"] * (1.0 + 2.2e-16 * rng.choice([-1, 0, 1], st["pt"].shape)))
    pert = OD.run(g, npz, fl, dp0, st2, bdt)
    return {n: P.rel_rms(pert[n], ref[n]) for n in ("u", "v", "w", "delp", "pt", "zh", "delz", "mfx", "mfy", "cx", "cy")}


def thin_akbk(bd, npz, delp):
    """an (ak, bk) whose 1 % thresholds (mix_dp, dyn_core.F90:2140) sit in the middle of each layer's range of the state: about
    half of the cells of every layer are `thin`, chains of them included"""
    v = bd.view(delp, "A", bd.is_, bd.ie, bd.js, bd.je)
    dref = 100.0 * 0.5 * (v.min(axis=(0, 1)) + v.max(axis=(0, 1)))
    return np.zeros(npz + 1), np.concatenate(([0.0], np.cumsum(dref))) / 1.0e5


def check_substeps(lib, nx=24, ny=16, npz=8, n_split=2, bdt=4.0, flags=None, tol=None, ic=None, pfull=None, ks=0, do_diss_est=False, akbk=None):
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    if do_diss_est:     # flagstruct%do_diss_est: a member of the gridstruct the context uploads
        import dataclasses
        g = dataclasses.replace(g, do_diss_est=True, prevent_diss_cooling=False)
    st, dp0 = make_state(bd, npz)
    apply_ic(bd, npz, st, ic)
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, **(flags or {}))
    if akbk == "thin":
        akbk = thin_akbk(bd, npz, st["delp"])
    ref = OD.run(g, npz, fl, dp0, st, bdt, pfull=pfull, ks=ks, akbk=akbk)
    if fl.fill_dp:          # mix_dp really acts in this run
        import dataclasses
        assert P.rel_rms(OD.run(g, npz, dataclasses.replace(fl, fill_dp=False), dp0, st, bdt)["delp"], ref["delp"]) > 1e-6
    if pfull is not None:   # fast_tau_w_sec / RF_fast really change the step
        off = DynFlags(n_split=n_split, ptop=N.PTOP, **dict(flags or {}, fast_tau_w_sec=0.0, rf_fast=False))
        assert P.rel_rms(OD.run(g, npz, off, dp0, st, bdt)["w"], ref["w"]) > 1e-6
    ctx = Context(g, npz, lib=lib)
    try:
        dc = DynCore(ctx, fl, dp0, pfull=pfull, ks=ks, akbk=akbk)
        dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        dc.run(bdt)
        got = dc.get_state()
        # exp / log are the same IEEE operation sequence on both sides (include/fv3_math.h), so the GPU is held to the
        # bound of the host-emulation harness: no conditioning-floor allowance for w any more
        tol = tol or 1e-13
        tols = {n: tol for n in ("u", "v", "w", "delp", "pt", "zh", "delz", "mfx", "mfy", "cx", "cy")}
        r = (bd.is_, bd.ie, bd.js, bd.je)
        out = {}
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("w", "A", r), ("delp", "A", r), ("pt", "A", r), ("zh", "A", r)):
            out[n] = P.assert_close(n, bd.view(got[n], kind, *rr), bd.view(ref[n], kind, *rr), tols[n])
        out["delz"] = P.assert_close("delz", got["delz"], ref["delz"], tols["delz"])
        om_got, om_ref = bd.view(dc.d["omga"].download(), "A", *r), bd.view(ref["omga"], "A", *r)
        out["omga"] = P.assert_close("omga", om_got, om_ref, tol)
        for n in ("mfx", "mfy", "cx", "cy"):
            out[n] = P.assert_close(n, got[n], ref[n], tols[n])
        if do_diss_est:   # diss_est(i,j,k) summed over the acoustic substeps (dyn_core.F90:805-811)
            assert np.max(np.abs(bd.view(ref["diss_est"], "A", *r))) > 0.0
            out["diss_est"] = P.assert_close("diss_est", bd.view(dc.d["diss_est"].download(), "A", *r), bd.view(ref["diss_est"], "A", *r), tol)
        # sanity: the step did something and stayed sane
        assert np.all(ref["delz"] < 0) and np.max(np.abs(ref["w"])) < 50.0
    finally:
        ctx.close()
    return out


def _oracle_fill2d(g, npz, q, delp, which):
    """fill2D after tracer_2d (fv_dynamics.F90:542-556): qt = q delp area, its halo, the sign-change fluxes"""
    import oracle_lib as O
    bd = g.bd
    for iq in which:
        qi = np.asfortranarray(q[:, :, :, iq])
        qt = bd.zeros("A", npz)
        O.fill2d_mass(g, npz, qi, delp, qt)
        OD._fill(bd, qt, "A")
        O.fill2d_apply(g, npz, qt, delp, qi)
        q[:, :, :, iq] = qi


def oracle_fv_step_hydro(g, npz, fl, st, ak, bk, q, bdt, k_split, remap_par, last_step=0, fill2d=()):
    """hydrostatic k_split loop over the oracle: dyn_core -> tracer_2d -> Lagrangian_to_Eulerian"""
    import oracle_lib as O
    bd = g.bd
    mdt = bdt / float(k_split)
    cur = {k: st[k].copy(order="F") for k in ("u", "v", "delp", "pt", "phis")}
    nq = 0 if q is None else q.shape[3]
    q = None if q is None else q.copy(order="F")
    out = None
    for n_map in range(1, k_split + 1):
        dp1 = cur["delp"].copy(order="F")
        OD._fill(bd, dp1, "A")
        if nq and fl.inline_q:
            cur["q"] = q
        f = OD.run_hydrostatic(g, npz, fl, cur, mdt)
        if nq and fl.inline_q:
            q = f["q"]
        elif nq:
            O.tracer_2d(g, npz, nq, q, dp1, f["mfx"], f["mfy"], f["cx"], f["cy"], fl.hord_tr, 0, 0, 0.0)
            _oracle_fill2d(g, npz, q, f["delp"], fill2d if fl.hord_tr < 8 else ())
        rf = dict(ps=bd.zeros("A"), pe=f["pe"], delp=f["delp"], pkz=f["pkz"], pk=f["pk"], u=f["u"], v=f["v"],
                  pt=f["pt"], peln=f["peln"], omga=bd.zeros("A", npz))
        if nq:
            rf["q"] = q
        if remap_par.get("remap_te"):
            rf["hs"], rf["te"] = st["phis"], bd.zeros("A", npz)
        O.lagrangian_to_eulerian(g, npz, dict(remap_par, last_step=(int(last_step) if n_map == k_split else 0)), rf, ak, bk)
        cur = dict(u=rf["u"], v=rf["v"], delp=rf["delp"], pt=rf["pt"], phis=st["phis"])
        cur.update({n: f[n] for n in ("du", "dv") if n in f})               # dyn_core's saved arrays (dyn_core.F90:278-283)
        out = dict(cur, q=q, pkz=rf["pkz"], ps=rf["ps"], pe=rf["pe"], peln=rf["peln"], pk=rf["pk"], te=rf.get("te"))
    return out


def apply_ic(bd, npz, st, ic):
    if ic == "westward":
        # every horizontal wind reversed: the upwind neighbour is the other one in each stencil (the default states have
        # u > 0 everywhere, which once hid a wrong lane at the east end of the c_sw strips)
        st["u"][...] = -st["u"]
        st["v"][...] = -st["v"]
    if ic == "test_case_1":
        # the doubly periodic test_case = 1 of the reference's solo core (tools/test_cases.F90:4689-4711): u = v = 10,
        # pt = 1, phis = 0, delp = 1 on i, j in 1..4 and 0 elsewhere -- here on a background of 1 (a
        # layer without mass divides 0 by 0 in c_sw's ptc)
        from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
        st["u"][...] = 10.0
        st["v"][...] = 10.0
        st["pt"][...] = 1.0
        st["phis"][...] = 0.0
        st["delp"][...] = 1.0
        ng = bd.ng
        st["delp"][ng:ng + 4, ng:ng + 4, :] += 1.0
        for k in range(npz):
            periodic_fill(bd, st["delp"][:, :, k], "A")


def check_fv_step_hydrostatic(lib, nx=24, ny=16, npz=10, nq=2, k_split=2, n_split=2, bdt=8.0, ic=None, flags=None, remap_te=False, kord_tm=-8):
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = make_state(bd, npz)
    apply_ic(bd, npz, st, ic)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=True, **(flags or {}))
    rng = np.random.default_rng(5)
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,))) if nq else None
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, remap_te=remap_te, kord_tm=kord_tm)
        ref = oracle_fv_step_hydro(g, npz, fl, st, ak, bk, q, bdt, k_split, dict(fv.remap_par, remap_te=int(remap_te)))
        fv.dc.set_state(st["u"], st["v"], np.zeros_like(st["w"]), st["delp"], st["pt"], st["delz"], st["phis"])
        if nq:
            fv.set_tracers(q)
        fv.step(bdt)
        d = fv.dc.d
        tol = 1e-13
        r = (bd.is_, bd.ie, bd.js, bd.je)
        out = {}
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r), ("ps", "A", r)):
            got_n, ref_n = bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr)
            out[n] = P.assert_close(n, got_n, ref_n, tol)
        for n in ("pkz", "pk", "peln"):
            out[n] = P.assert_close(n, d[n].download(), ref[n], tol)
        if nq:
            got = d["q"].download()
            for iq in range(nq):
                out[f"q{iq}"] = P.assert_close(f"q{iq}", bd.view(got[:, :, :, iq], "A", *r),
                                               bd.view(ref["q"][:, :, :, iq], "A", *r), tol)
    finally:
        ctx.close()
    return out


def oracle_fv_step(g, npz, fl, dp0, st, ak, bk, q, bdt, k_split, remap_par, last_step=False, fill2d=(), pfull=None, ks=0):
    """Oracle-orchestrated k_split loop (fv_dynamics.F90:460-665): dyn_core -> tracer_2d -> Lagrangian_to_Eulerian."""
    import oracle_lib as O
    bd = g.bd
    mdt = bdt / float(k_split)
    cur = {k: v.copy(order="F") for k, v in st.items()}
    nq = 0 if q is None else q.shape[3]
    q = None if q is None else q.copy(order="F")
    out = None
    for n_map in range(1, k_split + 1):
        dp1 = cur["delp"].copy(order="F")
        OD._fill(bd, dp1, "A")
        if fl.use_cond:
            OD._fill(bd, cur["q_con"], "A")                                  # fv_dynamics.F90:464 / :487
        if fl.moist_kappa:
            OD._fill(bd, cur["cappa"], "A")                                  # :465 / :488
        if nq and fl.inline_q:
            cur["q"] = q
        f = OD.run(g, npz, fl, dp0, cur, mdt, pfull=pfull, ks=ks)
        if nq and fl.inline_q:
            q = f["q"]
        elif nq:
            O.tracer_2d(g, npz, nq, q, dp1, f["mfx"], f["mfy"], f["cx"], f["cy"], fl.hord_tr, 0, 0, 0.0)
            _oracle_fill2d(g, npz, q, f["delp"], fill2d if fl.hord_tr < 8 else ())
        rf = dict(ps=bd.zeros("A"), pe=f["pe"], delp=f["delp"], pkz=bd.zeros("CC", npz), pk=f["pk"], u=f["u"], v=f["v"],
                  w=f["w"], delz=f["delz"], pt=f["pt"], peln=f["peln"], omga=f["omga"], ws=f["ws"])
        if nq:
            rf["q"] = q
        for n in ("q_con", "cappa"):
            if n in f:
                rf[n] = f[n]
        if remap_par.get("remap_te"):
            rf["hs"], rf["te"] = st["phis"], bd.zeros("A", npz)
        O.lagrangian_to_eulerian(g, npz, dict(remap_par, last_step=(int(last_step) if n_map == k_split else 0)), rf, ak, bk)
        cur = dict(u=rf["u"], v=rf["v"], w=rf["w"], delp=rf["delp"], pt=rf["pt"], delz=rf["delz"], phis=st["phis"])
        cur.update({n: f[n] for n in ("du", "dv") if n in f})
        for n in ("q_con", "cappa"):
            if n in rf:
                cur[n] = rf[n]
        out = dict(cur, q=q, pkz=rf["pkz"], ps=rf["ps"], pe=rf["pe"], peln=rf["peln"], pk=rf["pk"], te=rf.get("te"))
    return out


def check_fv_cycle_moist(lib, nx=24, ny=16, npz=10, k_split=2, n_split=2, bdt=8.0, moist_kappa=True, flags=None):
    """A whole fv_dynamics call with use_cond (+ moist_kappa): T -> theta_m with moist_cv, q_con transported by d_sw and
    entering the Riemann solvers, moist remap, back to T on the last step.  Oracle side: the conversion in numpy, then
    the oracle-orchestrated loop."""
    import parity_remap as R
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, KAPPA, RDGAS
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = make_state(bd, npz)
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + ny))
    nq = 7
    rng = np.random.default_rng(8)
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,)))
    q[..., 0] *= 0.02
    q[..., 1:6] *= 0.002
    for iq in range(nq):
        for k in range(npz):
            from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
            periodic_fill(bd, q[:, :, k, iq], "A")
    mp = dict(R.MOIST6, sphum=1)
    zvir = 0.6077
    # a temperature consistent with the balanced theta state
    th = st["pt"]
    dpc, thc = st["delp"][c], th[c]
    gm = 1.0 / (1.0 - KAPPA)
    T = th.copy(order="F")
    T[c] = thc * np.exp(KAPPA * gm * np.log((-RDGAS / GRAV) * dpc * thc / st["delz"]))
    # ---- oracle side: fv_dynamics.F90:305-317 / :323-326 and :381-388 in numpy ----
    dp1 = zvir * q[c + (slice(None), 0)]
    q_con, cappa = bd.zeros("A", npz), bd.zeros("A", npz)
    cvm, qc = N.np_moist_cv(q[c], mp, CP_AIR - RDGAS)
    q_con[c] = qc
    if moist_kappa:
        cappa[c] = RDGAS / (RDGAS + cvm / (1.0 + dp1))
        pkz = O.fexp(cappa[c] * O.flog((-RDGAS / GRAV) * dpc * T[c] * (1.0 + dp1) * (1.0 - qc) / st["delz"]))
    else:
        pkz = O.fexp(KAPPA * O.flog((-RDGAS / GRAV) * dpc * T[c] * (1.0 + dp1) / st["delz"]))
    th2 = T.copy(order="F")
    th2[c] = T[c] * (1.0 + dp1) * (1.0 - qc) / pkz
    for k in range(npz):
        periodic_fill(bd, th2[:, :, k], "A")
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, use_cond=True, moist_kappa=moist_kappa, **(flags or {}))
    ost = dict(st, pt=th2, q_con=q_con)
    if moist_kappa:
        ost["cappa"] = cappa
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, adiabatic=False, moist=mp, c2l_ord=2)
        opar = dict(fv.remap_par, **dict(mp, moist_kappa=int(moist_kappa), use_cond=1))
        opar.pop("sphum")
        opar["sphum"] = 1
        ref = oracle_fv_step(g, npz, fl, dp_ref, ost, ak, bk, q, bdt, k_split, opar, last_step=True)
        if not moist_kappa:   # use_cond alone: the library is told q_con (the reference keeps it from the previous step)
            pass
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], T, st["delz"], st["phis"])
        fv.set_tracers(q)
        if not moist_kappa:
            fv.dc.d["q_con"].upload(q_con)
        fv.step_from_temperature(bdt)
        d = fv.dc.d
        out = {}
        for n, kind, tol in (("pt", "A", 1e-13), ("delp", "A", 1e-13), ("w", "A", 1e-13), ("u", "U", 1e-13)):
            rr = r if kind == "A" else (bd.is_, bd.ie, bd.js, bd.je + 1)
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), tol)
        got_q = d["q"].download()
        for iq in (0, 1, 5):
            out[f"q{iq}"] = P.assert_close(f"q{iq}", bd.view(got_q[:, :, :, iq], "A", *r),
                                           bd.view(ref["q"][:, :, :, iq], "A", *r), 1e-13)
        Tn = bd.view(d["pt"].download(), "A", *r)
        assert 150.0 < Tn.min() and Tn.max() < 400.0
    finally:
        ctx.close()
    return out


def check_fv_cycle_from_temperature(lib, nx=24, ny=16, npz=10, k_split=2, n_split=2, bdt=8.0, tau=0.0, consv_am=False, rf_fast=False):
    """A whole adiabatic fv_dynamics call: T -> theta_v (fv_dynamics.F90:284-399), k_split loop, last remap back to T.
    Oracle side: the same conversion in numpy, then the oracle-orchestrated loop with last_step on the final cycle."""
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import GRAV, KAPPA, RDGAS
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = make_state(bd, npz)
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    # a temperature field consistent with the balanced theta_v state: T = theta_v * pkz(theta_v state)
    th = st["pt"]
    dpc, thc = bd.view(st["delp"], "A", *r), bd.view(th, "A", *r)
    gm = 1.0 / (1.0 - KAPPA)
    pkz0 = np.exp(KAPPA * gm * np.log((-RDGAS / GRAV) * dpc * thc / st["delz"]))     # p**kappa of the theta state
    T = th.copy(order="F")
    T[ng:ng + nx, ng:ng + ny, :] = thc * pkz0
    # oracle side conversion (fv_dynamics.F90:323-329, :389-397 with dp1 = 0)
    Tc = bd.view(T, "A", *r)
    pkz = O.fexp(KAPPA * O.flog((-RDGAS / GRAV) * dpc * Tc * (1.0 + 0.0) / st["delz"]))
    th2 = T.copy(order="F")
    th2[ng:ng + nx, ng:ng + ny, :] = Tc * (1.0 + 0.0) / pkz
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, rf_fast=rf_fast)
    rf_cutoff = 0.5 * (ak[npz // 2] + bk[npz // 2] * 1.0e5)          # the upper half of the column is damped
    ost = dict(st, pt=th2)
    if tau > 0.0 and not rf_fast:    # (RF_fast: no Rayleigh_Friction here, Ray_fast after every acoustic substep instead -- :362, dyn_core.F90:1057)
        # Rayleigh_Friction between the pkz evaluation and the conversion (fv_dynamics.F90:323-326, :368-376, :389-397)
        from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
        ph = ak + bk * 1.0e5
        pfull = (ph[1:] - ph[:-1]) / np.log(ph[1:] / ph[:-1])
        rf, kmax = O.rayleigh_rf(npz, abs(bdt), tau, rf_cutoff, N.PTOP, pfull)
        assert 0 < kmax < npz
        o_u, o_v, o_w = st["u"].copy(order="F"), st["v"].copy(order="F"), st["w"].copy(order="F")
        o_T, o_dz = T.copy(order="F"), st["delz"].copy(order="F")
        o_ua, o_va, o_u2f = bd.zeros("A", npz), bd.zeros("A", npz), bd.zeros("A", kmax)
        O.rayleigh_u2f(g, kmax, False, o_u, o_v, o_w, o_ua, o_va, o_u2f)
        for k in range(kmax):
            periodic_fill(bd, o_u2f[:, :, k], "A")
        O.rayleigh_apply(g, kmax, True, False, fl.cp_air, fl.rdgas, N.PTOP, pfull, rf, o_u2f, o_T, o_dz, o_u, o_v, o_w)
        th3 = o_T.copy(order="F")
        th3[ng:ng + nx, ng:ng + ny, :] = bd.view(o_T, "A", *r) * (1.0 + 0.0) / pkz
        for a, kind in ((o_u, "U"), (o_v, "V"), (o_w, "A"), (th3, "A")):
            for k in range(npz):
                periodic_fill(bd, a[:, :, k], kind)
        ost = dict(st, u=o_u, v=o_v, w=o_w, pt=th3, delz=o_dz)
    ca = None
    if consv_am:   # flagstruct%consv_am (fv_dynamics.F90:358-361, :747-800) with a made-up grid: latitudes, l2c_u / l2c_v and zxg of no sphere
        rng = np.random.default_rng(31)
        ca = dict(coslat=np.asfortranarray(rng.uniform(0.2, 1.0, bd.shape("A"))), l2c_u=np.asfortranarray(rng.uniform(-1, 1, bd.shape("U"))),
                  l2c_v=np.asfortranarray(rng.uniform(-1, 1, bd.shape("V"))), zxg=rng.uniform(-1e-3, 1e-3, (nx, ny)), omega=7.292e-5)
        radius = 6.3712e6
        o_ua, o_va = bd.zeros("A", npz), bd.zeros("A", npz)
        O.c2l(g, npz, 2, st["u"], st["v"], o_ua, o_va)
        teq, mf, ps2 = bd.zeros("CC"), bd.zeros("CC"), bd.zeros("A")
        O.compute_aam(g, npz, radius, ca["omega"], 1.0 / fl.grav, N.PTOP, ca["coslat"], o_ua, st["delp"], teq, mf, ps2)
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=0, k_split=k_split, tau=tau, rf_cutoff=rf_cutoff, consv_am=ca)
        assert fv.fl.tau == tau and fv.fl.rf_cutoff == rf_cutoff        # ONE flagstruct%tau: the argument reaches dyn_core's flags
        ph_ = ak + bk * 1.0e5                                            # fv_dynamics.F90:254-262: pfull, ks as FvDynamics forms them
        pf_ = (ph_[1:] - ph_[:-1]) / np.log(ph_[1:] / ph_[:-1])
        ks_ = max(int(np.argmax(bk != 0.0)) - 1, 0)
        ref = oracle_fv_step(g, npz, fv.fl, dp_ref, ost, ak, bk, None, bdt, k_split, fv.remap_par, last_step=True, pfull=pf_, ks=ks_)
        if rf_fast:    # the damping is in the run at all (it was silently dropped when tau lived in two places)
            off = oracle_fv_step(g, npz, DynFlags(n_split=n_split, ptop=N.PTOP), dp_ref, ost, ak, bk, None, bdt, k_split, fv.remap_par, last_step=True)
            assert P.rel_rms(off["u"], ref["u"]) > 1e-7
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], T, st["delz"], st["phis"])
        fv.step_from_temperature(bdt)
        d = fv.dc.d
        emu = "hostemu" in lib.path
        out = {}
        if consv_am:
            from gfdl_atmos_cubed_sphere_amd.global_sum import g_sum
            O.c2l(g, npz, 2, ref["u"], ref["v"], o_ua, o_va)
            te, ps1 = bd.zeros("CC"), bd.zeros("A")
            O.compute_aam(g, npz, radius, ca["omega"], 1.0 / fl.grav, N.PTOP, ca["coslat"], o_ua, ref["delp"], te, mf, ps1)
            cc = lambda a: a[ng:ng + nx, ng:ng + ny]
            area = [np.asarray(g.m["area"])[ng:ng + nx, ng:ng + ny]]
            te_2d = te - teq + 0.5 * bdt * (cc(ps2) + cc(ps1)) * ca["zxg"]
            u00 = -radius * g_sum([te_2d], area) / g_sum([mf], area)
            assert abs(u00) > 1e-8
            O.consv_am_apply(g, npz, u00, ca["l2c_u"], ca["l2c_v"], ref["u"], ref["v"])
            # u00 is a difference of column integrals ~ r^2 omega dm: what the two sides' fields differ by (<= 1e-13) is amplified in it
            assert abs(fv.last_u00 - u00) <= 1e-7 * abs(u00), (fv.last_u00, u00)
            for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je))):
                out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), 1e-11)
            out["u00_rel"] = abs(fv.last_u00 - u00) / abs(u00)
            for n, refa in (("aam2", te), ("m_fac", mf)):
                out[n] = P.assert_close(n, d[n].download(), refa, 1e-12)
        # cubed_to_latlon at the end (fv_dynamics.F90:911): c2l_ord4 of the final winds after their halo update
        from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill as pfill
        fu, fvv = d["u"].download(), d["v"].download()
        for k in range(npz):
            pfill(bd, fu[:, :, k], "U")
            pfill(bd, fvv[:, :, k], "V")
        r_ua, r_va = bd.zeros("A", npz), bd.zeros("A", npz)
        O.c2l(g, npz, 4, fu, fvv, r_ua, r_va)
        for n, refa in (("ua", r_ua), ("va", r_va)):
            out[n] = P.assert_close(n, bd.view(d[n].download(), "A", *r), bd.view(refa, "A", *r), 1e-13)
        for n, kind, tol in (("pt", "A", 1e-13), ("delp", "A", 1e-13), ("w", "A", 1e-13)):
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *r), bd.view(ref[n], kind, *r), tol)
        Tn = bd.view(d["pt"].download(), "A", *r)
        assert 150.0 < Tn.min() and Tn.max() < 400.0      # it is a temperature again
    finally:
        ctx.close()
    return out


def check_fv_step(lib, nx=24, ny=16, npz=10, nq=2, k_split=2, n_split=2, bdt=8.0, flags=None, fill2d=(), q_range=(0.0, 1.0), remap_te=False,
                  kord_tm=-8):
    """Whole model step (k_split x [substeps, tracer_2d, remap]) library vs oracle."""
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = make_state(bd, npz)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, **(flags or {}))
    rng = np.random.default_rng(5)
    q = np.asfortranarray(rng.uniform(q_range[0], q_range[1], bd.shape("A", npz) + (nq,))) if nq else None
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, fill2d=fill2d, moist_phys=bool(fill2d), remap_te=remap_te, kord_tm=kord_tm)
        ref = oracle_fv_step(g, npz, fl, dp_ref, st, ak, bk, q, bdt, k_split, dict(fv.remap_par, remap_te=int(remap_te)), fill2d=fill2d)
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], st["pt"], st["delz"], st["phis"])
        if nq:
            fv.set_tracers(q)
        fv.step(bdt)
        d = fv.dc.d
        tol = 1e-13
        r = (bd.is_, bd.ie, bd.js, bd.je)
        out = {}
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("w", "A", r), ("delp", "A", r), ("pt", "A", r), ("ps", "A", r)):
            t = tol
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), t)
        for n in ("delz", "pkz", "pk", "peln"):
            out[n] = P.assert_close(n, d[n].download(), ref[n], tol)
        if nq:
            got = d["q"].download()
            for iq in range(nq):
                out[f"q{iq}"] = P.assert_close(f"q{iq}", bd.view(got[:, :, :, iq], "A", *r),
                                               bd.view(ref["q"][:, :, :, iq], "A", *r), tol)
    finally:
        ctx.close()
    return out


def check_substeps_hydrostatic(lib, nx=24, ny=16, npz=8, n_split=2, bdt=4.0, flags=None, ic=None, do_diss_est=False):
    """hydrostatic substep loop (c_sw, geopk, p_grad_c, d_sw, geopk, one_grad_p with external-mode damping)"""
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    if do_diss_est:
        import dataclasses
        g = dataclasses.replace(g, do_diss_est=True, prevent_diss_cooling=False)
    st, dp0 = make_state(bd, npz)
    apply_ic(bd, npz, st, ic)
    hst = {k: st[k] for k in ("u", "v", "delp", "pt", "phis")}
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=True, **(flags or {}))
    akbk = thin_akbk(bd, npz, st["delp"]) if fl.fill_dp else None
    ref = OD.run_hydrostatic(g, npz, fl, hst, bdt, akbk=akbk)
    if fl.fill_dp:
        import dataclasses
        assert P.rel_rms(OD.run_hydrostatic(g, npz, dataclasses.replace(fl, fill_dp=False), hst, bdt)["delp"], ref["delp"]) > 1e-6
    ctx = Context(g, npz, lib=lib)
    try:
        dc = DynCore(ctx, fl, dp0, akbk=akbk)
        z = np.zeros_like(st["w"])
        dc.set_state(st["u"], st["v"], z, st["delp"], st["pt"], st["delz"], st["phis"])
        dc.run(bdt)
        d = dc.d
        tol = 1e-13
        r = (bd.is_, bd.ie, bd.js, bd.je)
        out = {}
        for n, kind, rr in (("u", "U", (bd.is_, bd.ie, bd.js, bd.je + 1)), ("v", "V", (bd.is_, bd.ie + 1, bd.js, bd.je)),
                            ("delp", "A", r), ("pt", "A", r)):
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), tol)
        for n in ("mfx", "mfy", "cx", "cy", "pk", "pkz", "peln"):
            out[n] = P.assert_close(n, d[n].download(), ref[n], tol)
        out["pe"] = P.assert_close("pe", d["pe"].download()[1:-1, :, 1:-1], ref["pe"][1:-1, :, 1:-1], tol)
        if do_diss_est:
            assert np.max(np.abs(bd.view(ref["diss_est"], "A", *r))) > 0.0
            out["diss_est"] = P.assert_close("diss_est", bd.view(d["diss_est"].download(), "A", *r), bd.view(ref["diss_est"], "A", *r), tol)
    finally:
        ctx.close()
    return out


def check_fv_cycle_consv(lib, nx=24, ny=16, npz=10, k_split=2, n_split=2, bdt=8.0, hydrostatic=False, consv_te=1.0, adiabatic=False,
                         remap_te=False):
    """consv_te: compute_total_energy before the loop (fv_dynamics.F90:345), the energy fixer of the last remap (fv_mapz.F90:643-772:
    te_2d, zsum, the two reproducing global sums, dtmp) and step 9a with dtmp (:793-821), a whole fv_dynamics call from T.
    Oracle side: the oracle's restatements with math.fsum for the area-weighted sums (the exact sum, which the EFP sum of the host
    reproduces to the last rounding)."""
    import math
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.lib import CP_AIR, GRAV, KAPPA, RDGAS
    from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
    bd = Bounds(1, nx, 1, ny)
    g = P.make_grid(bd, False)
    st, dp0 = make_state(bd, npz)
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + ny))
    nq = 2
    rng = np.random.default_rng(12)
    q = np.asfortranarray(rng.uniform(0.0, 1.0, bd.shape("A", npz) + (nq,)))
    q[..., 0] *= 0.015
    for iq in range(nq):
        for k in range(npz):
            periodic_fill(bd, q[:, :, k, iq], "A")
    zvir = 0.0 if adiabatic else 0.6077
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.5
    ak, bk = N.PTOP * (1.0 - sig), sig.copy()
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    fl = DynFlags(n_split=n_split, ptop=N.PTOP, hydrostatic=hydrostatic)
    dpc = st["delp"][c]
    th = st["pt"]
    # a temperature consistent with the theta state, and the pressure arrays p_var would have left (hydrostatic)
    pe3 = N.PTOP + np.concatenate([np.zeros((nx, ny, 1)), np.cumsum(dpc, axis=2)], axis=2)
    peln3 = O.flog(pe3).reshape(pe3.shape)
    pk3 = O.fexp(KAPPA * peln3).reshape(pe3.shape)
    if hydrostatic:
        pkz = (pk3[:, :, 1:] - pk3[:, :, :-1]) / (KAPPA * (peln3[:, :, 1:] - peln3[:, :, :-1]))
    else:
        gm = 1.0 / (1.0 - KAPPA)
        pkz = np.exp(KAPPA * gm * np.log((-RDGAS / GRAV) * dpc * th[c] / st["delz"]))
    T = th.copy(order="F")
    T[c] = th[c] * pkz / (1.0 + zvir * q[c + (slice(None), 0)])
    pe = np.zeros((nx + 2, npz + 1, ny + 2), order="F")
    pe[1:-1, :, 1:-1] = np.transpose(pe3, (0, 2, 1))
    peln = np.asfortranarray(np.transpose(peln3, (0, 2, 1)))
    ctx = Context(g, npz, lib=lib)
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=nq, k_split=k_split, adiabatic=adiabatic, c2l_ord=2, consv_te=consv_te, remap_te=remap_te)
        par = dict(fv.remap_par, remap_te=int(remap_te))
        # ---- oracle side ----
        area = np.asarray(g.m["area"])[c]
        te0 = bd.zeros("CC")
        if consv_te > 0:
            O.compute_total_energy(g, npz, par, False, st["u"], st["v"], None if hydrostatic else st["w"],
                                   None if hydrostatic else st["delz"], T, st["delp"], q, None, pe if hydrostatic else None,
                                   peln if hydrostatic else None, st["phis"], te0)
        dp1 = zvir * q[c + (slice(None), 0)]
        if hydrostatic:
            pkz_o = np.asfortranarray(pkz)
        else:
            pkz_o = O.fexp(KAPPA * O.flog((-RDGAS / GRAV) * dpc * T[c] * (1.0 + dp1) / st["delz"])).reshape(dpc.shape)
        th2 = T.copy(order="F")
        th2[c] = T[c] * (1.0 + dp1) / pkz_o
        for k in range(npz):
            periodic_fill(bd, th2[:, :, k], "A")
        ost = dict(st, pt=th2)
        if hydrostatic:
            ref = oracle_fv_step_hydro(g, npz, fl, ost, ak, bk, q, bdt, k_split, par, last_step=2)
        else:
            ref = oracle_fv_step(g, npz, fl, dp_ref, ost, ak, bk, q, bdt, k_split, par, last_step=2)
        te2, z1, z0 = bd.zeros("CC"), bd.zeros("CC"), bd.zeros("CC")
        O.energy_fixer_sums(g, npz, dict(par, te=ref["te"]) if remap_te else par, consv_te < 0, ref["u"], ref["v"], ref.get("w"),
                            ref.get("delz"), ref["pt"], ref["delp"], ref["q"], ref["pe"], ref["peln"], st["phis"], ref["pkz"], ref["pk"],
                            te0, te2, z1, z0)
        zs = math.fsum(((z0 if hydrostatic else z1) * area).ravel())
        if consv_te < 0:
            dtmp = consv_te * (GRAV * bdt * 4.0 * np.pi * 6.3712e6 ** 2) / zs
        else:
            dtmp = consv_te * math.fsum((te2 * area).ravel()) / zs
        O.remap_finish(g, npz, par, dtmp, ref["pt"], ref["pkz"], ref["q"])
        if consv_te == 1.0 and not (adiabatic and not hydrostatic) and not remap_te:
            # the point of it: the energy of the final state is the initial one again (to the linearisation of the fixer), while the
            # unfixed step lost / gained `drift`
            te_end = bd.zeros("CC")
            O.compute_total_energy(g, npz, par, False, ref["u"], ref["v"], ref.get("w"), ref.get("delz"), ref["pt"], ref["delp"], ref["q"],
                                   None, ref["pe"], ref["peln"], st["phis"], te_end)
            e0, e1 = math.fsum((te0 * area).ravel()), math.fsum((te_end * area).ravel())
            drift = math.fsum((te2 * area).ravel())
            assert abs(e1 - e0) < 0.05 * abs(drift), (e0, e1, drift)
        # ---- library ----
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], T, st["delz"], st["phis"])
        fv.set_tracers(q)
        if hydrostatic:
            fv.dc.d["pe"].upload(pe)
            fv.dc.d["peln"].upload(peln)
            fv.dc.d["pkz"].upload(np.asfortranarray(pkz))
        fv.step_from_temperature(bdt)
        d = fv.dc.d
        out = {"dtmp": abs(fv.dtmp - dtmp) / abs(dtmp)}
        assert out["dtmp"] < 1e-12, (fv.dtmp, dtmp)
        assert abs(dtmp) > 1e-12                              # the fixer did something
        names = (("pt", "A"), ("delp", "A"), ("u", "U")) + (() if hydrostatic else (("w", "A"),))
        for n, kind in names:
            rr = r if kind == "A" else (bd.is_, bd.ie, bd.js, bd.je + 1)
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), 1e-13)
        if consv_te > 0:
            out["te_2d"] = P.assert_close("te_2d", d["te_2d"].download(), te2, 1e-9)     # a difference of two large sums
            out["te0_2d"] = P.assert_close("te0_2d", d["te0_2d"].download(), te0, 1e-14)
    finally:
        ctx.close()
    return out


def check_supercell_step(lib, nx=48, ny=32, npz=32, k_split=1, n_split=3, bdt=9.0, dxy=1000.0, flags=None):
    """BASELINE config 4's initial condition in small: the doubly periodic supercell (test_case = 17, tools/test_cases.F90:4966-5057:
    Weisman-Klemp sounding, sheared wind, warm bubble, water vapour as tracer 1) through a whole nonhydrostatic fv_dynamics call
    (T -> theta_v with the virtual effect, k_split x (n_split substeps, tracer_2d, remap), back to T) against the oracle loop"""
    from gfdl_atmos_cubed_sphere_amd.fv_dynamics import FvDynamics
    from gfdl_atmos_cubed_sphere_amd.grid import doubly_periodic
    from gfdl_atmos_cubed_sphere_amd.layout import periodic_fill
    from gfdl_atmos_cubed_sphere_amd.lib import GRAV, KAPPA, RDGAS
    from gfdl_atmos_cubed_sphere_amd.test_cases import supercell
    bd = Bounds(1, nx, 1, ny)
    g = doubly_periodic(bd, nx + 1, ny + 1, dx_const=dxy, dy_const=dxy)
    sig = np.linspace(0.0, 1.0, npz + 1) ** 1.3
    ptop = 2000.0
    ak, bk = ptop * (1.0 - sig), sig.copy()
    st = supercell(bd, npz, ak, bk, dxy, dxy, dt_rad=8.0 * dxy)
    q = st.pop("q")
    for n, kind in (("u", "U"), ("v", "V"), ("w", "A"), ("delp", "A"), ("pt", "A")):
        for k in range(npz):
            periodic_fill(bd, st[n][:, :, k], kind)
    for k in range(npz):
        periodic_fill(bd, q[:, :, k, 0], "A")
    r = (bd.is_, bd.ie, bd.js, bd.je)
    ng = bd.ng
    c = (slice(ng, ng + nx), slice(ng, ng + ny))
    zvir = 0.6077
    dp_ref = (ak[1:] - ak[:-1]) + (bk[1:] - bk[:-1]) * 1.0e5
    fl = DynFlags(n_split=n_split, ptop=ptop, **(flags or {}))
    T = st["pt"]
    assert T[c].max() - np.median(T[c][:, :, -4]) > 0.5        # the bubble is there
    dp1 = zvir * q[c + (slice(None), 0)]
    pkz = O.fexp(KAPPA * O.flog((-RDGAS / GRAV) * st["delp"][c] * T[c] * (1.0 + dp1) / st["delz"])).reshape(dp1.shape)
    th = T.copy(order="F")
    th[c] = T[c] * (1.0 + dp1) / pkz
    for k in range(npz):
        periodic_fill(bd, th[:, :, k], "A")
    ctx = Context(g, npz, lib=lib)
    out = {}
    try:
        fv = FvDynamics(ctx, fl, ak, bk, nq=1, k_split=k_split, adiabatic=False, c2l_ord=2)
        ref = oracle_fv_step(g, npz, fl, dp_ref, dict(st, pt=th), ak, bk, q, bdt, k_split, dict(fv.remap_par), last_step=True)
        fv.dc.set_state(st["u"], st["v"], st["w"], st["delp"], T, st["delz"], st["phis"])
        fv.set_tracers(q)
        fv.step_from_temperature(bdt)
        d = fv.dc.d
        for n, kind in (("pt", "A"), ("delp", "A"), ("w", "A"), ("u", "U")):
            rr = r if kind == "A" else (bd.is_, bd.ie, bd.js, bd.je + 1)
            out[n] = P.assert_close(n, bd.view(d[n].download(), kind, *rr), bd.view(ref[n], kind, *rr), 1e-12)
        out["q"] = P.assert_close("q", bd.view(d["q"].download()[:, :, :, 0], "A", *r), bd.view(ref["q"][:, :, :, 0], "A", *r), 1e-12)
        w = bd.view(d["w"].download(), "A", *r)
        assert np.all(np.isfinite(w)) and np.max(np.abs(w)) > 1e-4          # the bubble rises
        area = dxy * dxy
        m0, m1 = np.sum(st["delp"][c]) * area, np.sum(bd.view(d["delp"].download(), "A", *r)) * area
        assert abs(m1 - m0) <= 1e-13 * m0                                   # the air mass of the periodic domain
    finally:
        ctx.close()
    return out
