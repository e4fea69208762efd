"""The acoustic substep loop (model/dyn_core.F90:313-1286, nonhydrostatic, non-nested, grid_type=4)
orchestrated over the ORACLE's routines with numpy arrays and in-place reference semantics.  Test
infrastructure: the CPU side of the whole-substep parity tests."""
from __future__ import annotations

import numpy as np

import oracle_lib as O
from gfdl_atmos_cubed_sphere_amd.dyn_core import DynFlags, level_coefficients
from gfdl_atmos_cubed_sphere_amd.layout import Bounds, periodic_fill


def _fill(bd, a, kind):
    if a.ndim == 2:
        periodic_fill(bd, a, kind)
    elif a.ndim == 4:
        for n in range(a.shape[3]):
            _fill(bd, a[:, :, :, n], kind)
    else:
        for k in range(a.shape[2]):
            periodic_fill(bd, a[:, :, k], kind)


def run(g, npz: int, fl: DynFlags, dp_ref, st: dict, bdt: float, pfull=None, ks=0, akbk=None):
    """st: u, v, w, delp, pt (halo'd), delz (CC x npz), phis (A).  Returns the updated state dict.
    pfull, ks: for fast_tau_w_sec > 0 / RF_fast (the profiles are evaluated as on the reference's first call of a run that starts here)"""
    rff = None
    if fl.fast_tau_w_sec > 1.0e-5:
        rff = O.fast_tau_w_rff(npz, 0.5 * bdt / float(fl.n_split), fl.fast_tau_w_sec, fl.rf_cutoff, fl.ptop, pfull)
    O.set_fast_tau_w(rff)
    try:
        return _run(g, npz, fl, dp_ref, st, bdt, pfull, ks, akbk)
    finally:
        O.set_fast_tau_w(None)


def _run(g, npz: int, fl: DynFlags, dp_ref, st: dict, bdt: float, pfull=None, ks=0, akbk=None):
    bd: Bounds = g.bd
    f = {k: np.asfortranarray(v.copy()) for k, v in st.items()}
    nx, ny = bd.nx, bd.ny
    for n, kind, nk in (("delpc", "A", npz), ("ptc", "A", npz), ("uc", "V", npz), ("vc", "U", npz), ("ua", "A", npz),
                        ("va", "A", npz), ("omga", "A", npz), ("ut", "A", npz), ("vt", "A", npz), ("divgd", "B", npz),
                        ("gz", "A", npz + 1), ("pkc", "A", npz + 1), ("zh", "A", npz + 1), ("pk3", "A", npz + 1),
                        ("crx", "CX", npz), ("xfx", "CX", npz), ("cry", "CY", npz), ("yfx", "CY", npz),
                        ("mfx", "FX", npz), ("mfy", "FY", npz), ("cx", "CX", npz), ("cy", "CY", npz),
                        ("heat_s", "CC", npz), ("diss_e", "CC", npz), ("pk", "CC", npz + 1)):
        f[n] = bd.zeros(kind, nk)
    f["ws3"], f["ws"] = bd.zeros("A"), bd.zeros("CC")
    f["pe"] = np.zeros((nx + 2, npz + 1, ny + 2), order="F")
    f["peln"] = np.zeros((nx, npz + 1, ny), order="F")
    zs = np.asfortranarray(f["phis"] * (1.0 / fl.grav))
    lev = level_coefficients(npz, fl)
    cn = dict(grav=fl.grav, rdgas=fl.rdgas, cp_air=fl.cp_air, akap=fl.akap, ptop=fl.ptop, p_fac=fl.p_fac, a_imp=fl.a_imp)
    n_split = fl.n_split
    dt = bdt / float(n_split)
    dt2, rdt = 0.5 * dt, 1.0 / dt
    ptk, peln1 = fl.ptop ** fl.akap, np.log(fl.ptop)
    par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm,
               hord_dp=fl.hord_dp, nord=1, nord_v=1, nord_w=1, nord_t=1, dddmp=fl.dddmp, d2_bg=0.0, d4_bg=fl.d4_bg,
               damp_v=0.0, damp_w=0.0, damp_t=0.0, d_con=0.0, kgb=fl.ke_bg, hydrostatic=0, use_cond=int(fl.use_cond))
    qc = lambda: f["q_con"] if fl.use_cond else None
    cap = lambda: f["cappa"] if fl.moist_kappa else None
    ndif = np.concatenate([lev["nord_v"], lev["nord_v"][-1:]]).astype(np.int32)
    damp = np.concatenate([lev["damp_vt"], lev["damp_vt"][-1:]])
    _fill(bd, f["delp"], "A"); _fill(bd, f["pt"], "A"); _fill(bd, f["u"], "U"); _fill(bd, f["v"], "V")
    heating = fl.d_con > 1.0e-5
    f["heat_source"] = bd.zeros("A", npz)
    f["pkz"] = bd.zeros("CC", npz)
    for it in range(1, n_split + 1):
        remap_step = it == n_split
        _fill(bd, f["w"], "A")
        if it == 1:
            gz = f["gz"]
            i0, j0 = bd.ng, bd.ng
            gz[i0:i0 + nx, j0:j0 + ny, npz] = zs[i0:i0 + nx, j0:j0 + ny]
            for k in range(npz - 1, -1, -1):
                gz[i0:i0 + nx, j0:j0 + ny, k] = gz[i0:i0 + nx, j0:j0 + ny, k + 1] - f["delz"][:, :, k]
            _fill(bd, gz, "A")
            f["zh"][...] = gz
        else:
            f["gz"][...] = f["zh"]
        cs = dict(delpc=f["delpc"], delp=f["delp"], ptc=f["ptc"], pt=f["pt"], u=f["u"], v=f["v"], w=f["w"], uc=f["uc"],
                  vc=f["vc"], ua=f["ua"], va=f["va"], wc=f["omga"], ut=f["ut"], vt=f["vt"], divg_d=f["divgd"])
        O.c_sw_3d(g, npz, cs, nord=fl.nord, dt2=dt2, hydrostatic=False)
        if fl.nord > 0:
            _fill(bd, f["divgd"], "B")
        O.update_dz_c(g, npz, dt2, dp_ref, zs, f["ut"], f["vt"], f["gz"], f["ws3"])
        O.riem_solver_c(g, npz, dt2, cn, f["phis"], f["omga"], f["ptc"], f["delpc"], f["gz"], f["pkc"], f["ws3"], qc(), cap())
        O.p_grad_c(g, npz, dt2, f["delpc"], f["pkc"], f["gz"], f["uc"], f["vc"], False)
        _fill(bd, f["uc"], "V"); _fill(bd, f["vc"], "U")
        delp_start = f["delp"].copy(order="F")
        ds = dict(delpc=f["vt"], delp=f["delp"], ptc=f["ptc"], pt=f["pt"], u=f["u"], v=f["v"], w=f["w"], uc=f["uc"],
                  vc=f["vc"], ua=f["ua"], va=f["va"], divg_d=f["divgd"], mfx=f["mfx"], mfy=f["mfy"], cx=f["cx"],
                  cy=f["cy"], crx=f["crx"], cry=f["cry"], xfx=f["xfx"], yfx=f["yfx"], heat_source=f["heat_s"],
                  diss_est=f["diss_e"])
        if fl.use_cond:
            ds["q_con"] = f["q_con"]
        if fl.inline_q and "q" in f:                                       # dyn_core.F90:341 / :573, sw_core.F90:1020-1043
            _fill(bd, f["q"], "A")
            ds["inline_q"] = f["q"]
        O.d_sw_3d(g, npz, par, lev, ds)
        if fl.use_cond:
            _fill(bd, f["q_con"], "A")                                       # dyn_core.F90:825 / :852
        if heating:                                                        # dyn_core.F90:798-803
            i0, j0 = bd.ng, bd.ng
            f["heat_source"][i0:i0 + nx, j0:j0 + ny, :] += f["heat_s"]
        if g.do_diss_est:                                                  # dyn_core.F90:805-811 (zero on the first call, :285)
            f.setdefault("diss_est", bd.zeros("A", npz))
            f["diss_est"][bd.ng:bd.ng + nx, bd.ng:bd.ng + ny, :] += f["diss_e"]
        if fl.beta < -0.1:                                                 # dyn_core.F90:745-747, :791-848 (zeros when d_ext = 0)
            f.setdefault("divg2", bd.zeros("A"))
            O.divg2_ext(g, npz, fl.d_ext, delp_start, f["vt"], f["divg2"])
        if fl.fill_dp:                                                     # dyn_core.F90:820
            O.mix_dp(g, npz, False, akbk[0], akbk[1], f["w"], f["delp"], f["pt"])
        _fill(bd, f["delp"], "A"); _fill(bd, f["pt"], "A")
        O.update_dz_d(g, npz, ndif.copy(), damp.copy(), fl.hord_tm, dp_ref, zs, f["zh"], f["crx"], f["cry"], f["xfx"],
                      f["yfx"], f["ws"], rdt)
        O.riem_solver3(g, npz, dt, cn, zs, f["w"], f["delz"], f["pt"], f["delp"], f["zh"], f["pe"], f["pkc"], f["pk3"],
                       f["pk"], f["peln"], f["ws"], fl.use_logp, remap_step, fl.beta < -0.1, qc(), cap())   # fp_out: dyn_core.F90:939
        _fill(bd, f["zh"], "A"); _fill(bd, f["pkc"], "A")
        if remap_step:
            O.pe_halo(g, npz, fl.ptop, f["pe"], f["delp"])
        O.pk3_halo(g, npz, fl.ptop, fl.akap, f["pk3"], f["delp"], fl.use_logp)
        i0, i1, j0, j1 = bd.ng - 2, bd.ng + nx + 2, bd.ng - 2, bd.ng + ny + 2
        f["gz"][i0:i1, j0:j1, :] = f["zh"][i0:i1, j0:j1, :] * fl.grav
        if fl.beta > 0.0:                                                  # dyn_core.F90:1027-1028, beta_d :398-406
            for n, kind in (("du", "U"), ("dv", "V")):
                f.setdefault(n, bd.zeros(kind, npz))
            O.split_p_grad(g, npz, f["u"], f["v"], f["pkc"], f["gz"], f["delp"], f["pk3"], 0.0 if it == 1 else fl.beta, dt,
                           peln1 if fl.use_logp else ptk, f["du"], f["dv"])
        elif fl.beta < -0.1:                                               # dyn_core.F90:1029-1030
            O.one_grad_p_nh(g, npz, dt, fl.ptop, f["divg2"], f["u"], f["v"], f["pkc"], f["gz"], f["delp"])
        else:
            O.nh_p_grad(g, npz, f["u"], f["v"], f["pkc"], f["gz"], f["delp"], f["pk3"], dt, peln1 if fl.use_logp else ptk)
        if fl.rf_fast and fl.tau > 0.0:                                    # dyn_core.F90:1057-1060
            kmax, k_rf, _, rf = O.ray_fast_profile(npz, ks, abs(dt), fl.tau, fl.rf_cutoff, fl.ptop, pfull, dp_ref)
            O.ray_fast(g, npz, kmax, k_rf, rf, dp_ref, False, f["u"], f["v"], f["w"])
        if it != n_split:
            _fill(bd, f["u"], "U"); _fill(bd, f["v"], "V")
        elif fl.use_old_omega:                                             # dyn_core.F90:409-421, :1182-1191
            i0, j0 = bd.ng, bd.ng
            pem = fl.ptop + np.cumsum(delp_start[i0:i0 + nx, j0:j0 + ny, :], axis=2)
            pe_c = np.transpose(f["pe"][1:-1, 1:, 1:-1], (0, 2, 1))
            f["omga"][i0:i0 + nx, j0:j0 + ny, :] = (pe_c - pem) * rdt
    # dissipative heating (dyn_core.F90:296-308, :1300-1355)
    if fl.convert_ke or (fl.do_vort_damp and fl.vtdm4 > 1.0e-4):
        n_con = npz
    elif fl.d2_bg_k1 < 1.0e-3:
        n_con = 0
    else:
        n_con = 1 if fl.d2_bg_k2 < 1.0e-3 else 2
    if n_con != 0 and heating:
        _fill(bd, f["heat_source"], "A")
        O.del2_cubed(g, npz, 0.20 * g.da_min, min(3, fl.nord + 1), f["heat_source"])
        O.apply_heat_source(g, npz, n_con, False, bdt, fl.delt_max, fl.cp_air, fl.cp_air - fl.rdgas, fl.rdgas, fl.grav,
                            f["pt"], f["heat_source"], f["delp"], f["delz"], f["pkz"],
                            f["cappa"] if fl.moist_kappa else None)                      # :1338-1340
    return f


def run_hydrostatic(g, npz: int, fl: DynFlags, st: dict, bdt: float, akbk=None):
    """hydrostatic branch of the substep loop over the oracle's routines.  st: u, v, delp, pt (halo'd), phis."""
    bd: Bounds = g.bd
    f = {k: np.asfortranarray(v.copy()) for k, v in st.items()}
    nx, ny = bd.nx, bd.ny
    for n, kind, nk in (("delpc", "A", npz), ("ptc", "A", npz), ("uc", "V", npz), ("vc", "U", npz), ("ua", "A", npz),
                        ("va", "A", npz), ("ut", "A", npz), ("vt", "A", npz), ("divgd", "B", npz),
                        ("gz", "A", npz + 1), ("pkc", "A", npz + 1), ("crx", "CX", npz), ("xfx", "CX", npz),
                        ("cry", "CY", npz), ("yfx", "CY", npz), ("mfx", "FX", npz), ("mfy", "FY", npz), ("cx", "CX", npz),
                        ("cy", "CY", npz), ("heat_s", "CC", npz), ("diss_e", "CC", npz), ("pk", "CC", npz + 1),
                        ("pkz", "CC", npz), ("heat_source", "A", npz)):
        f[n] = bd.zeros(kind, nk)
    f["divg2"] = bd.zeros("A")
    f["pe"] = np.zeros((nx + 2, npz + 1, ny + 2), order="F")
    f["peln"] = np.zeros((nx, npz + 1, ny), order="F")
    lev = level_coefficients(npz, fl)
    n_split = fl.n_split
    dt = bdt / float(n_split)
    dt2 = 0.5 * dt
    ptk = fl.ptop ** fl.akap
    par = dict(dt=dt, hord_tr=fl.hord_tr, hord_mt=fl.hord_mt, hord_vt=fl.hord_vt, hord_tm=fl.hord_tm,
               hord_dp=fl.hord_dp, nord=1, nord_v=1, nord_w=1, nord_t=1, dddmp=fl.dddmp, d2_bg=0.0, d4_bg=fl.d4_bg,
               damp_v=0.0, damp_w=0.0, damp_t=0.0, d_con=0.0, kgb=fl.ke_bg, hydrostatic=1, use_cond=0)
    heating = fl.d_con > 1.0e-5
    _fill(bd, f["delp"], "A"); _fill(bd, f["pt"], "A"); _fill(bd, f["u"], "U"); _fill(bd, f["v"], "V")
    i0, j0 = bd.ng, bd.ng
    for it in range(1, n_split + 1):
        remap_step = it == n_split
        cs = dict(delpc=f["delpc"], delp=f["delp"], ptc=f["ptc"], pt=f["pt"], u=f["u"], v=f["v"], uc=f["uc"], vc=f["vc"],
                  ua=f["ua"], va=f["va"], ut=f["ut"], vt=f["vt"], divg_d=f["divgd"])
        O.c_sw_3d(g, npz, cs, nord=fl.nord, dt2=dt2, hydrostatic=True)
        if fl.nord > 0:
            _fill(bd, f["divgd"], "B")
        O.geopk(g, npz, fl.ptop, fl.akap, fl.cp_air, f["pe"], f["peln"], f["delpc"], f["pkc"], f["gz"], f["phis"], f["ptc"],
                f["pkz"], True)
        O.p_grad_c(g, npz, dt2, f["delpc"], f["pkc"], f["gz"], f["uc"], f["vc"], True)
        _fill(bd, f["uc"], "V"); _fill(bd, f["vc"], "U")
        delp_old = f["delp"].copy(order="F")
        ds = dict(delpc=f["vt"], delp=f["delp"], ptc=f["ptc"], pt=f["pt"], u=f["u"], v=f["v"], uc=f["uc"], vc=f["vc"],
                  ua=f["ua"], va=f["va"], divg_d=f["divgd"], mfx=f["mfx"], mfy=f["mfy"], cx=f["cx"], cy=f["cy"],
                  crx=f["crx"], cry=f["cry"], xfx=f["xfx"], yfx=f["yfx"], heat_source=f["heat_s"], diss_est=f["diss_e"])
        if fl.inline_q and "q" in f:
            _fill(bd, f["q"], "A")
            ds["inline_q"] = f["q"]
        O.d_sw_3d(g, npz, par, lev, ds)
        if heating:
            f["heat_source"][i0:i0 + nx, j0:j0 + ny, :] += f["heat_s"]
        if g.do_diss_est:                                                  # dyn_core.F90:805-811
            f.setdefault("diss_est", bd.zeros("A", npz))
            f["diss_est"][bd.ng:bd.ng + nx, bd.ng:bd.ng + ny, :] += f["diss_e"]
        O.divg2_ext(g, npz, fl.d_ext, delp_old, f["vt"], f["divg2"])
        if fl.fill_dp:                                                     # dyn_core.F90:820
            O.mix_dp(g, npz, True, akbk[0], akbk[1], None, f["delp"], f["pt"])
        _fill(bd, f["delp"], "A"); _fill(bd, f["pt"], "A")
        O.geopk(g, npz, fl.ptop, fl.akap, fl.cp_air, f["pe"], f["peln"], f["delp"], f["pkc"], f["gz"], f["phis"], f["pt"],
                f["pkz"], False)
        if remap_step:
            f["pk"][...] = f["pkc"][i0:i0 + nx, j0:j0 + ny, :]
        if fl.beta > 0.0:                                                  # dyn_core.F90:1018-1019
            for n, kind in (("du", "U"), ("dv", "V")):
                f.setdefault(n, bd.zeros(kind, npz))
            O.grad1_p_update(g, npz, f["divg2"], f["u"], f["v"], f["pkc"], f["gz"], dt, ptk, 0.0 if it == 1 else fl.beta, f["du"], f["dv"])
        else:
            O.one_grad_p_hydro(g, npz, dt, ptk, f["divg2"], f["u"], f["v"], f["pkc"], f["gz"])
        if it != n_split:
            _fill(bd, f["u"], "U"); _fill(bd, f["v"], "V")
    if fl.convert_ke or (fl.do_vort_damp and fl.vtdm4 > 1.0e-4):
        n_con = npz
    elif fl.d2_bg_k1 < 1.0e-3:
        n_con = 0
    else:
        n_con = 1 if fl.d2_bg_k2 < 1.0e-3 else 2
    if n_con != 0 and heating:
        _fill(bd, f["heat_source"], "A")
        O.del2_cubed(g, npz, 0.20 * g.da_min, min(3, fl.nord + 1), f["heat_source"])
        O.apply_heat_source(g, npz, n_con, True, bdt, fl.delt_max, fl.cp_air, fl.cp_air - fl.rdgas, fl.rdgas, fl.grav,
                            f["pt"], f["heat_source"], f["delp"], f["pkz"], f["pkz"])
    return f
